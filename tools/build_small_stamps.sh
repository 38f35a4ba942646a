#!/bin/bash
# Scratch build of the library with phase stamps inside small_step_kernel (tools/small_step_probe.py reads them):
#   bash tools/build_small_stamps.sh && OIVAE_LIB_PATH=$PWD/oi_vae_amd/liboivae_hip_stamps.so python tools/small_step_probe.py
set -e
cd "$(dirname "$0")/../oi_vae_amd/csrc"
make -s
/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -fPIC --offload-arch=gfx950 -std=c++17 -ffp-contract=off -DOIVAE_SMALL_STAMPS -c small_step.hip -o build/small_step_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v -e small_step -e waittest) build/small_step_stamps.o -o ../liboivae_hip_stamps.so
