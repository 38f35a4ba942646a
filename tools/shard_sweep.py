#!/usr/bin/env python3
"""Per-rank compute time of the scale-up step at the shard sizes a data-parallel
run gives each GPU (global batch 65536 over 1/2/4/8 ranks), measured on ONE GPU.
The multi-GPU bench itself is the driver's to launch; this shows what the strong
scaling curve is made of before the all-reduce is added.

usage: python tools/shard_sweep.py [rows ...]     (default 65536 32768 16384 8192)
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
  rows_list = [int(a) for a in sys.argv[1:]] or [65536, 32768, 16384, 8192]
  device = torch.device("cuda", 0)
  torch.cuda.set_device(device)
  for rows in rows_list:
    eng, X = bench.init_engine(device, rows, seed=100)
    egen = torch.Generator(device=device).manual_seed(7)
    ev = []

    def step(timed):
      eps = torch.randn(1, rows, bench.K, generator=egen, device=device)
      eng.trace = ev if timed else None
      eng.forward_backward(X, eps, 1.0, lam=bench.LAM)
      eng.trace = None
      eng.adam_step(bench.LR)
      eng.sgd_prox_step(bench.WS_LR, bench.WS_LR * bench.LAM)

    for _ in range(3):
      step(False)
    torch.cuda.synchronize()
    steps = 10
    t0 = time.perf_counter()
    for _ in range(steps):
      step(True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    per = {}
    for name, a, b in ev:
      per.setdefault(name, []).append(a.elapsed_time(b))
    print(json.dumps({"rows": rows, "ms_per_step": ms, "samples_per_s_per_gpu": rows / ms * 1e3,
                      "ideal_ms_at_1gpu_rate": None,
                      "entry_ms": {k: sum(v) / len(v) for k, v in per.items()}}), flush=True)
    del eng, X
    torch.cuda.empty_cache()


if __name__ == "__main__":
  main()
