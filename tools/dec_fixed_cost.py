#!/usr/bin/env python3
"""Fixed (row-independent) cost of the decoder launch at G=1024: time at a few tiny batches."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_scale import _engine
from oi_vae_amd import _lib
eng = _engine(1024, 64, 64, 64, 64, seed=3)
g = torch.Generator(device="cuda").manual_seed(1)
for B in (128, 512, 2048, 8192):
  X = torch.rand(B, 65536, generator=g, device="cuda")
  eps = torch.randn(1, B, 64, generator=g, device="cuda")
  for m in ("x3", "v4"):
    _lib.set_test_hooks(OIVAE_DEC=m)
    eng.forward_backward(X, eps, 1.0)
    eng.trace = []
    for _ in range(5):
      eng.forward_backward(X, eps, 1.0)
    torch.cuda.synchronize()
    t = {}
    for name, a, b in eng.trace:
      t.setdefault(name, []).append(a.elapsed_time(b))
    eng.trace = None
    print(B, m, {k: round(float(np.mean(v)), 3) for k, v in t.items()})
