#!/usr/bin/env python3
"""What the per-entry-point HIP events of Engine.trace cost: ms per step at one shard size with and without them."""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402

dev = torch.device("cuda", 0)
for rows in [int(a) for a in sys.argv[1:]] or [8192, 65536]:
  eng, X = bench.init_engine(dev, rows, seed=100)
  egen = torch.Generator(device=dev).manual_seed(7)

  def step(trace):
    eps = torch.randn(1, rows, bench.K, generator=egen, device=dev)
    eng.trace = [] if trace else None
    eng.forward_backward(X, eps, 1.0, lam=bench.LAM)
    eng.trace = None
    eng.adam_step(bench.LR)
    eng.sgd_prox_step(bench.WS_LR, bench.WS_LR * bench.LAM)

  out = {"rows": rows}
  for rep in range(2):
    for trace in (True, False):
      for _ in range(3):
        step(trace)
      torch.cuda.synchronize()
      n = 40
      t0 = time.perf_counter()
      for _ in range(n):
        step(trace)
      torch.cuda.synchronize()
      out["%s_%d" % ("events" if trace else "plain", rep)] = round((time.perf_counter() - t0) / n * 1e3, 4)
  print(json.dumps(out), flush=True)
  del eng, X
  torch.cuda.empty_cache()
