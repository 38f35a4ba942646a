#!/bin/bash
# Three separate rocprofv3 --pmc passes over a short bench.py run (counters never
# share a run with tracing; each block's slot limits respected: FETCH_SIZE uses 3 of
# the 4 TCC slots, WRITE_SIZE 2, at most 8 SQ counters).  Run on the MI355X box from
# the repo root:   bash tools/pmc_collect.sh gpurun_out/pmc_rNN
# then fold with   python3 tools/pmc_summarize.py profiles/rNN_pmc_per_kernel.json gpurun_out/pmc_rNN/*
set -e
OUT=${1:-gpurun_out/pmc}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$ROOT/$OUT/fetch" -- python3 $ARGS > "$ROOT/$OUT/fetch.log" 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$ROOT/$OUT/write" -- python3 $ARGS > "$ROOT/$OUT/write.log" 2>&1
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d "$ROOT/$OUT/sq" -- python3 $ARGS > "$ROOT/$OUT/sq.log" 2>&1
echo pmc passes done
