#!/usr/bin/env python3
"""dec_x3_64 (OIVAE_DEC=x3), dec_x3_64p (OIVAE_DEC=p) and dec_x3_64g (OIVAE_DEC=g), dec_x3_64s (OIVAE_DEC=s) beside
dec_mfma64_v4: agreement on a ragged small case, then time at the scale-up shape."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_scale import _engine
from oi_vae_amd import _lib  # noqa: E402


def setmode(m):
  _lib.set_test_hooks(OIVAE_DEC=m)


def agree():
  eng = _engine(24, 64, 64, 64, 64, seed=1)
  g = torch.Generator(device="cuda").manual_seed(5)
  B, S = 200, 2
  X = torch.rand(B, 24 * 64, generator=g, device="cuda")
  eps = torch.randn(S, B, 64, generator=g, device="cuda")
  res = {}
  for m in ("v4", "x3", "p", "g", "s"):
    setmode(m)
    out = eng.forward_backward(X, eps, 0.7)
    torch.cuda.synchronize()
    res[m] = (eng.flat_grads.clone(), out["scalars"].clone())
  a, b = res[os.environ.get("CHECK_MODE", "g")], res["v4"]
  print("scalars:   ", a[1][:6].tolist())
  print("scalars v4:", b[1][:6].tolist())
  for name in eng.seg_off:
    o, n = eng.seg_off[name], eng.seg_numel[name]
    if n:
      ga, gb = a[0][o:o + n], b[0][o:o + n]
      print("%-10s rel err %.3e" % (name, ((ga - gb).abs().max() / gb.abs().max().clamp(min=1e-30)).item()))


def timing():
  eng = _engine(1024, 64, 64, 64, 64, seed=3)
  g = torch.Generator(device="cuda").manual_seed(1)
  B = 65536
  X = torch.rand(B, 65536, generator=g, device="cuda")
  eps = torch.randn(1, B, 64, generator=g, device="cuda")
  for m in ("g", "s", "g", "s"):
    setmode(m)
    eng.forward_backward(X, eps, 1.0)
    eng.trace = []
    for _ in range(3):
      eng.forward_backward(X, eps, 1.0)
    torch.cuda.synchronize()
    t = {}
    for name, a, b in eng.trace:
      t.setdefault(name, []).append(a.elapsed_time(b))
    eng.trace = None
    print(m, {k: round(float(np.mean(v)), 3) for k, v in t.items()})


if __name__ == "__main__":
  agree()
  if len(sys.argv) > 1:
    timing()
