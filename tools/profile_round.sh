set -e
R=gpurun_out/${1:-r02c}
mkdir -p $R
ROOT=$(pwd)
timeout -k 10 400 python3 bench.py > $R/bench_line.json 2> $R/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$R/prof -o bench -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $ROOT/$R/bench_line_under_rocprof.json 2> $ROOT/$R/rocprof.err
cd $ROOT
bash tools/pmc_collect.sh $R/pmc
timeout -k 10 300 python3 tools/shard_sweep.py > $R/shard_sweep.jsonl 2> $R/shard.err
timeout -k 10 300 python3 tools/config_steps.py > $R/config_steps.jsonl 2> $R/config.err
timeout -k 10 300 python3 tools/mixed_dims.py > $R/mixed_dims.jsonl 2> $R/mixed.err
ls $R $R/prof | head -40
