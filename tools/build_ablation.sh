#!/bin/bash
# Scratch build of the library with -DOIVAE_ABL_<TAG> (several tags joined by '+') on the decoder translation units
# (timing only; see tools/dec_ablate.py).
set -euo pipefail
TAGS="$1"
HERE="$(cd "$(dirname "$0")/.." && pwd)"
C="$HERE/oi_vae_amd/csrc"
mkdir -p "$HERE/tools/probe"
DEFS=""
for t in ${TAGS//+/ }; do DEFS="$DEFS -DOIVAE_ABL_$t"; done
/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -fPIC --offload-arch=gfx950 -std=c++17 $DEFS -c "$C/mfma_decoder.hip" -o "$HERE/tools/probe/abl_$TAGS.o"
/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -fPIC --offload-arch=gfx950 -std=c++17 $DEFS -c "$C/xcheck_decoders.hip" -o "$HERE/tools/probe/ablx_$TAGS.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls "$C"/build/*.o | grep -v -e mfma_decoder -e xcheck_decoders -e waittest -e stamps) \
  "$HERE/tools/probe/abl_$TAGS.o" "$HERE/tools/probe/ablx_$TAGS.o" -o "$HERE/tools/probe/liboivae_abl_$TAGS.so"
rm -f "$HERE/tools/probe/abl_$TAGS.o" "$HERE/tools/probe/ablx_$TAGS.o"
