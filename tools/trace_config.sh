#!/bin/bash
# rocprofv3 kernel trace of one small configuration's step (tools/config_steps.py NAME): gpurun_out/prof_cfg/
NAME=${1:-MEG}
R=$(pwd)/gpurun_out/prof_cfg; mkdir -p $R
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R -o s -- python3 /root/repo/tools/config_steps.py "$NAME" > $R/out.json 2> $R/err.log
cat $R/out.json | cut -c1-400
