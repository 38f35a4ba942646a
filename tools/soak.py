#!/usr/bin/env python3
"""Soak run of the scale-up step: SECONDS of back-to-back steps (default 240), alternating batches, then the library's
error words (a bounded LDS wait that ran out would have raised them), finiteness of every parameter and the step
rate over the run in windows -- evidence that the hand-over protocol of the decoder does not stall or time out under
sustained load.   usage: python tools/soak.py [seconds]"""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
eng, X = bench.init_engine(dev, bench.GLOBAL_BATCH, seed=100)
X2 = torch.rand(bench.GLOBAL_BATCH, bench.G * bench.DG, generator=torch.Generator(device=dev).manual_seed(900), device=dev)
egen = torch.Generator(device=dev).manual_seed(7)
windows, n, t_start = [], 0, time.perf_counter()
while time.perf_counter() - t_start < seconds:
  t0 = time.perf_counter()
  for i in range(200):
    eps = torch.randn(1, bench.GLOBAL_BATCH, bench.K, generator=egen, device=dev)
    eng.forward_backward(X if i % 2 == 0 else X2, eps, 1.0, lam=bench.LAM)
    eng.adam_step(bench.LR)
    eng.sgd_prox_step(bench.WS_LR, bench.WS_LR * bench.LAM)
  torch.cuda.synchronize()
  eng.check_errors()
  windows.append((time.perf_counter() - t0) / 200 * 1e3)
  n += 200
  print("steps %6d  window %.3f ms/step  elbo %.1f" % (n, windows[-1], float(eng.scalars[5])), flush=True)
finite = bool(torch.isfinite(eng.flat_params).all())
print(json.dumps({"steps": n, "seconds": time.perf_counter() - t_start, "ms_per_step_first_window": windows[0],
                  "ms_per_step_last_window": windows[-1], "ms_per_step_max_window": max(windows),
                  "error_words": "clear", "parameters_finite": finite,
                  "zero_rows_of_Ws": int((eng.p["Ws"] == 0).all(-1).sum())}))
assert finite
