#!/usr/bin/env python3
"""Two-layer generator shapes OUTSIDE {32, 64}^4 (any K, H, M <= 64, uniform D_g <= 64, a multiple of 4): decoder entry
point on the default path (the all-64 fp16-term kernel on zero-padded copies, csrc/dec_pad64.hip, from kPadMinWork
per-group work on) against the shape-generic kernels, at a large and a small problem.  This is the measurement behind
kPadMinWork (csrc/common.h).   usage: python tools/narrow_dims.py"""
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oi_vae_amd import _lib  # noqa: E402
from tests.test_gpu_scale import _engine  # noqa: E402

SHAPES = [(8, 8, 8, 8), (16, 8, 16, 16), (16, 16, 16, 16), (10, 8, 24, 64), (24, 24, 24, 24), (16, 16, 48, 40),
          (40, 24, 56, 44), (48, 48, 48, 48), (64, 64, 64, 48), (33, 17, 64, 12)]
for (G, B) in ((256, 16384), (8, 256)):
  for (K, H, M, dg) in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.rand(B, G * dg, generator=g, device="cuda")
    eps = torch.randn(1, B, K, generator=g, device="cuda")
    row = {"G": G, "B": B, "K": K, "H": H, "M": M, "D_g": dg, "work": K * H + H * M + M * dg}
    for tag, env in (("default_ms", {}), ("generic_ms", {"OIVAE_FORCE_GENERIC": "1"})):
      _lib.set_test_hooks(**env)
      eng = _engine(G, dg, K, H, M, seed=3)
      import ctypes
      row[tag.replace("_ms", "_layout")] = int(_lib.load().oivae_dz_layout(ctypes.byref(eng._shape), B, 1))
      ev = []
      for i in range(6):
        eng.trace = ev if i >= 2 else None
        eng.forward_backward(X, eps, 1.0)
      torch.cuda.synchronize()
      t = [a.elapsed_time(b) for n, a, b in ev if n == "decoder"]
      row[tag] = round(sum(t) / len(t), 4)
      _lib.set_test_hooks(**{k: None for k in env})
    print(json.dumps(row), flush=True)
