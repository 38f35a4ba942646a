#!/bin/bash
# A/B of two builds of the library on ONE box, alternating: tools/ab_libs.sh <other .so> [rows ...]
# (the shipped build is oi_vae_amd/liboivae_hip.so; OIVAE_LIB_PATH selects the other one)
set -e
other=$1; shift
rows=${@:-65536 8192}
for i in 1 2; do
  echo "== shipped"; timeout -k 10 200 python tools/shard_sweep.py $rows 2>&1 | grep rows
  echo "== $other"; OIVAE_LIB_PATH=$other timeout -k 10 200 python tools/shard_sweep.py $rows 2>&1 | grep rows
done
