#!/bin/bash
# rocprofv3 kernel trace of the step at one shard size (default 8192 rows): gpurun_out/prof_shard/
ROWS=${1:-8192}
R=$(pwd)/gpurun_out/prof_shard; mkdir -p $R
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R -o s -- python3 /root/repo/tools/shard_sweep.py $ROWS > $R/out.json 2> $R/err.log
cat $R/out.json | cut -c1-300
