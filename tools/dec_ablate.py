#!/usr/bin/env python3
"""Timing-only ablations of the scale-up decoder kernel: the shipped library beside a scratch build with one
cost compiled out (results of the scratch build are wrong by construction).

  make -C oi_vae_amd/csrc                      # the shipped library
  tools/build_ablation.sh NOATOMIC             # -> tools/probe/liboivae_abl_NOATOMIC.so
  python tools/dec_ablate.py [NOATOMIC ...]    # one child process per library

Each child prints the decoder entry point's time at 65536 and 8192 rows.
Tags (mfma_decoder.hip, OIVAE_ABL_<TAG>): NOATOMIC, NOCMFMA, NOWMFMA, NOCSPLIT, NOWSPLIT, NOPUB, NOTANH, NOX, CHAIN2, TIMES,
C16 / W16 (the chain / weight-gradient wave's matrix flops issued as 16x16x32 instead of 32x32x16 instructions on the
same operand registers: what the instruction shape alone is worth at the power limit).
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(tag):
  sys.path.insert(0, ROOT)
  import torch
  from oi_vae_amd import _lib
  if tag != "shipped" and not tag.startswith("shipped"):
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "liboivae_abl_%s.so" % tag)
  import bench
  dev = torch.device("cuda", 0)
  for rows in (65536, 8192):
    eng, X = bench.init_engine(dev, rows, seed=100)
    egen = torch.Generator(device=dev).manual_seed(7)
    ev = []
    for i in range(8):
      eps = torch.randn(1, rows, bench.K, generator=egen, device=dev)
      eng.trace = ev if i >= 3 else None
      eng.forward_backward(X, eps, 1.0, lam=bench.LAM)
    torch.cuda.synchronize()
    per = {}
    for name, a, b in ev:
      per.setdefault(name, []).append(a.elapsed_time(b))
    print(json.dumps({"lib": tag, "rows": rows, "entry_ms": {k: round(sum(v) / len(v), 4) for k, v in per.items()}}), flush=True)
    del eng, X
    torch.cuda.empty_cache()


if __name__ == "__main__":
  if len(sys.argv) > 2 and sys.argv[1] == "--child":
    child(sys.argv[2])
  else:
    for tag in ["shipped"] + sys.argv[1:]:
      rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--child", tag])
      if rc != 0:
        sys.exit(rc)
