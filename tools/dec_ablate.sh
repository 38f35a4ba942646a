#!/bin/bash
# Diagnostic: time the v3 decoder with one class of non-MFMA work removed at a time
# (template parameter ABL of dec_mfma64_v3).  Build here (CPU container):
#   bash tools/dec_ablate.sh build
# run on the MI355X box from the repo root:
#   bash tools/dec_ablate.sh run > gpurun_out/ablate.log
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/tools/diag
if [ "$1" = build ]; then
  mkdir -p $OUT
  C="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wno-unused-result -DOIVAE_DIAG_VARIANTS"
  S=$ROOT/oi_vae_amd/csrc
  /opt/rocm/bin/hipcc $C -ffp-contract=off -c $S/prox.hip -o $OUT/prox.o
  /opt/rocm/bin/hipcc $C -DOIVAE_RSQRT14_BIN="\"$ROOT/oi_vae_amd/data/rsqrt14_u16.bin\"" -c $S/api.hip -o $OUT/api.o
  for f in generic mfma_decoder mfma_encoder; do /opt/rocm/bin/hipcc $C -c $S/$f.hip -o $OUT/$f.o; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/*.o -o $OUT/liboivae_diag.so
  rm -f $OUT/*.o
  exit 0
fi
export OIVAE_HIP_LIB=$OUT/liboivae_diag.so
for abl in 0 1 2 4 8 16 32 64 127; do
  echo -n "ABL=$abl "
  OIVAE_DEC_ABLATE=$abl timeout -k 10 120 python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('decoder_ms', round(d['entry_ms']['decoder'],3))"
done
