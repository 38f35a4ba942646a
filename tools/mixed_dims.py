#!/usr/bin/env python3
"""Decoder entry point time of two-layer generator shapes with mixed 32/64 dimensions.  Default path: D_g = 64 shapes run
the all-64 fp16-term kernel on zero-padded copies (csrc/dec_pad64.hip), H = M = D_g = 32 shapes the same kernel on
PAIRS of groups, the other D_g = 32 shapes dec_mfma<K, H, M, DG> (fp32 matrix instruction); OIVAE_DEC=v1 forces dec_mfma for every shape, OIVAE_FORCE_GENERIC the shape-generic kernels.
Per shape: ms of the decoder entry point and ms per algorithmic TFLOP (3 G N (2KH + 2HM + 2M D_g)) relative to the
all-64 shape's.   usage: python tools/mixed_dims.py"""
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oi_vae_amd import _lib  # noqa: E402
from tests.test_gpu_scale import _engine  # noqa: E402

G, B = 256, 16384
base = None
for (K, H, M, dg) in [(64, 64, 64, 64), (32, 64, 64, 64), (64, 32, 64, 64), (64, 64, 32, 64), (32, 32, 32, 64), (64, 32, 64, 32),
                     (32, 32, 32, 32), (64, 32, 32, 32), (64, 64, 64, 32), (32, 64, 32, 32)]:
  g = torch.Generator(device="cuda").manual_seed(1)
  X = torch.rand(B, G * dg, generator=g, device="cuda")
  eps = torch.randn(1, B, K, generator=g, device="cuda")
  row = {"G": G, "B": B, "K": K, "H": H, "M": M, "D_g": dg}
  for tag, env in (("default_ms", {}), ("fp32_instruction_ms", {"OIVAE_DEC": "v1"}), ("generic_ms", {"OIVAE_FORCE_GENERIC": "1"})):
    if tag == "generic_ms" and (K, H, M, dg) == (64, 64, 64, 64):
      continue
    _lib.set_test_hooks(**env)
    eng = _engine(G, dg, K, H, M, seed=3)
    ev = []
    for i in range(6):
      eng.trace = ev if i >= 2 else None
      eng.forward_backward(X, eps, 1.0)
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) for n, a, b in ev if n == "decoder"]
    row[tag] = round(sum(t) / len(t), 3)
    _lib.set_test_hooks(**{k: None for k in env})
  flops = 3 * G * B * (2 * K * H + 2 * H * M + 2 * M * dg)
  row["ms_per_algorithmic_tflop"] = round(row["default_ms"] / (flops / 1e12), 4)
  if base is None:
    base = row["ms_per_algorithmic_tflop"]
  row["per_flop_vs_all64"] = round(row["ms_per_algorithmic_tflop"] / base, 2)
  print(json.dumps(row), flush=True)
