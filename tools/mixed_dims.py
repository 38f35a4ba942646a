#!/usr/bin/env python3
"""Decoder entry point time of a two-layer generator shape with mixed 32/64 dimensions: dec_mfma<K, H, M, DG> (fp32 matrix
instruction) against the shape-generic kernels (OIVAE_FORCE_GENERIC).  usage: python tools/mixed_dims.py"""
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oi_vae_amd import _lib  # noqa: E402
from tests.test_gpu_scale import _engine  # noqa: E402

G, B = 256, 16384
for (K, H, M, dg) in [(64, 32, 64, 32), (32, 64, 64, 64), (64, 64, 32, 64)]:
  g = torch.Generator(device="cuda").manual_seed(1)
  X = torch.rand(B, G * dg, generator=g, device="cuda")
  eps = torch.randn(1, B, K, generator=g, device="cuda")
  row = {"G": G, "B": B, "K": K, "H": H, "M": M, "D_g": dg}
  for tag, hook in (("matrix_ms", None), ("generic_ms", "1")):
    _lib.set_test_hooks(OIVAE_FORCE_GENERIC=hook)
    eng = _engine(G, dg, K, H, M, seed=3)
    ev = []
    for i in range(6):
      eng.trace = ev if i >= 2 else None
      eng.forward_backward(X, eps, 1.0)
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) for n, a, b in ev if n == "decoder"]
    row[tag] = round(sum(t) / len(t), 3)
  _lib.set_test_hooks(OIVAE_FORCE_GENERIC=None)
  print(json.dumps(row), flush=True)
