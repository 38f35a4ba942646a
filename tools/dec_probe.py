#!/usr/bin/env python3
"""Where a dec_x3_64g workgroup's time goes: wall-clock stamps written by a scratch build of the decoder
(tools/build_ablation.sh TIMES -> tools/probe/liboivae_abl_TIMES.so), read back after one launch.

usage: python tools/dec_probe.py [rows ...]        (default 8192 65536)
Stamps of wave 0 (a chain wave): 0 start, 1/2 after the set-up of the pair's first/second group, 3 row loop starts,
4 this wave's row loop done, 5 all waves' row loops done, 6 end.
"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oi_vae_amd import _lib  # noqa: E402

SETUP = "--setup" in sys.argv   # tools/build_ablation.sh TIMES2: the sub-phases of the set-up of a pair's second group
if SETUP:
  sys.argv.remove("--setup")
END = "--end" in sys.argv       # tools/build_ablation.sh TIMES3: the sub-phases of the end of the kernel, second group
if END:
  sys.argv.remove("--end")
_lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "liboivae_abl_TIMES3.so" if END else ("liboivae_abl_TIMES2.so" if SETUP else "liboivae_abl_TIMES.so"))
import bench  # noqa: E402


def main():
  rows_list = [int(a) for a in sys.argv[1:]] or [8192, 65536]
  dev = torch.device("cuda", 0)
  raw = ctypes.CDLL(_lib.LIB_PATH)
  raw.oivae_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
  for rows in rows_list:
    eng, X = bench.init_engine(dev, rows, seed=100)
    egen = torch.Generator(device=dev).manual_seed(7)
    for _ in range(4):
      eps = torch.randn(1, rows, bench.K, generator=egen, device=dev)
      eng.forward_backward(X, eps, 1.0, lam=bench.LAM)
    torch.cuda.synchronize()
    nwg = 512
    buf = np.zeros(nwg * 8, dtype=np.uint64)
    rc = raw.oivae_probe_read(buf.ctypes.data, nwg * 8)
    assert rc == 0, rc
    t = buf.reshape(nwg, 8).astype(np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0     # 100 MHz
    if SETUP or END:
      names = (["matrices and vectors into LDS", "A = W1 Ws^T", "maxima and gmu bounds", "da1 bounds", "scales", "images"] if SETUP else
               ["dW2 partials into LDS", "dW2 summed and stored", "P partials into LDS", "P summed, W1 and Ws into LDS", "dW1 and dWs products", "bias and sigma gradients"])
      d = np.diff(us[:, :7], axis=1)
      print(json.dumps({"rows": rows, "second_group_us_median": {n: round(float(np.median(d[:, i])), 2) for i, n in enumerate(names)},
                        "total": round(float(np.median(us[:, 6] - us[:, 0])), 2)}), flush=True)
      del eng, X
      torch.cuda.empty_cache()
      continue
    names = ["setup g0", "setup g1", "barrier", "row loop (wave 0)", "wait for the other waves", "end of kernel"]
    d = np.diff(us[:, :7], axis=1)
    first = us[:, 0] < np.median(us[:, 0])    # the first round of workgroups
    out = {"rows": rows, "launch_us": float(us[:, 6].max()),
           "start_us_median_round1": float(np.median(us[first, 0])), "start_us_median_round2": float(np.median(us[~first, 0])),
           "phase_us_median": {n: round(float(np.median(d[:, i])), 2) for i, n in enumerate(names)},
           "phase_us_max": {n: round(float(d[:, i].max()), 2) for i, n in enumerate(names)},
           "workgroup_us_median": float(np.median(us[:, 6] - us[:, 0]))}
    print(json.dumps(out), flush=True)
    del eng, X
    torch.cuda.empty_cache()


if __name__ == "__main__":
  main()
