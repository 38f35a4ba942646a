#!/usr/bin/env python3
"""Timing variants of the wide-prox launch (G=8192, K=256, H=8): with/without the mask and norm outputs, with/without
the threshold arithmetic, and a plain torch copy of the same bytes for reference.  Diagnostic only."""
import ctypes
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oi_vae_amd import _lib  # noqa: E402

if len(sys.argv) > 1:   # another build of the library (a scratch build with other -DOIVAE_PROX_R8 / _R64)
  _lib.LIB_PATH = sys.argv[1]
L = _lib.init_device(0)
Gp, Kp, Hp = 8192, 256, 8
gen = torch.Generator(device="cuda").manual_seed(3)
W0 = torch.randn(Gp, Kp, Hp, generator=gen, device="cuda")
dW = torch.randn(Gp, Kp, Hp, generator=gen, device="cuda")
t = float(torch.sqrt((W0 ** 2).sum(-1)).median())
mask = torch.zeros(Gp, Kp, dtype=torch.uint8, device="cuda")
norms = torch.zeros(Gp, Kp, dtype=torch.float32, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, reps=20):
  Ws = [W0.clone() for _ in range(reps + 2)]
  evs = []
  for W in Ws:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(W); b.record()
    evs.append((a, b))
  torch.cuda.synchronize()
  ms = sorted(a.elapsed_time(b) for a, b in evs[2:])
  return sum(ms) / len(ms) * 1e3


nbytes = 12 * Gp * Kp * Hp
for name, fn in [
  ("sgd+prox, mask+norms", lambda W: L.oivae_sgd_prox_f32(W.data_ptr(), dW.data_ptr(), Gp * Kp, Hp, 1e-4, t, 1, mask.data_ptr(), norms.data_ptr(), st)),
  ("sgd+prox, no outputs", lambda W: L.oivae_sgd_prox_f32(W.data_ptr(), dW.data_ptr(), Gp * Kp, Hp, 1e-4, t, 1, None, None, st)),
  ("sgd only (do_prox=0)", lambda W: L.oivae_sgd_prox_f32(W.data_ptr(), dW.data_ptr(), Gp * Kp, Hp, 1e-4, t, 0, None, None, st)),
  ("prox only (8 B/elt)", lambda W: L.oivae_prox_f32(W.data_ptr(), Gp * Kp, Hp, t, None, None, st)),
  ("torch addcmul (12 B/elt)", lambda W: W.add_(dW, alpha=-1e-4)),
]:
  us = timeit(fn)
  b = nbytes if "8 B" not in name else 8 * Gp * Kp * Hp
  print("%-28s %7.1f us  %6.0f GB/s" % (name, us, b / us * 1e-3))
