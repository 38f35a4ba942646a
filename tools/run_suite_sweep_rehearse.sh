#!/bin/bash
# One GPU call: the GPU test suite, the shard sweep at 65536 / 8192 rows and the two-rank rehearsal of bench.py's
# N > 1 path (both ranks on cuda:0 over gloo).  Outputs under gpurun_out/<tag>_*.
TAG=${1:-r2g}
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t_$TAG.log 2>&1
rc=$?
tail -5 gpurun_out/t_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/shard_sweep.py 65536 8192 > gpurun_out/shard_$TAG.jsonl 2> gpurun_out/shard_$TAG.err || exit $?
cut -c1-400 gpurun_out/shard_$TAG.jsonl
OIVAE_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 2 > gpurun_out/rehearse_$TAG.json 2> gpurun_out/rehearse_$TAG.err || exit $?
python - <<PY
import json
d = json.loads(open("gpurun_out/rehearse_$TAG.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d.get("comm"))
PY
