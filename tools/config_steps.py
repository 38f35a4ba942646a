#!/usr/bin/env python3
"""Full-step time of the small / odd BASELINE configurations on one GPU (launch- and latency-bound shapes that
run through the shape-generic kernels): bars, mocap, MEG-like, wide-prox.  Prints steps/s per configuration."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oi_vae_amd import Engine, ModelSpec  # noqa: E402

MOCAP = [3, 3, 3, 3, 3, 3, 3, 2, 3, 1, 1, 2, 1, 2, 2, 3, 1, 1, 2, 1, 2, 3, 1, 2, 1, 3, 1, 2, 1]
rng = np.random.default_rng(0)
MEG_DIMS = rng.integers(40, 260, 44).tolist()
MEG_DIMS[-1] += (-sum(MEG_DIMS)) % 4   # D a multiple of 4: rows of X 16-byte aligned (the MFMA encoder's requirement)
CONFIGS = {
  "bars (G=8x8, K=16, H=1, B=64)": dict(dims=[8] * 8, K=16, H=1, typ="linear", M=0, B=64, S=1),
  "mocap (G=29 ragged, K=8, H=8, B=64, S=10)": dict(dims=MOCAP, K=8, H=8, typ="tanh_linear", M=0, B=64, S=10),
  "MEG-like (G=44 ragged ~150, K=10, H=1, B=4096)": dict(dims=MEG_DIMS, K=10, H=1, typ="linear", M=0, B=4096, S=1),
  "wide-prox (G=8192x4, K=256, H=8, B=1024)": dict(dims=[4] * 8192, K=256, H=8, typ="linear", M=0, B=1024, S=1),
}
dev = torch.device("cuda", 0)
only = sys.argv[1:]   # substrings of configuration names; none = all
for name, c in CONFIGS.items():
  if only and not any(o in name for o in only):
    continue
  D = sum(c["dims"])
  eng = Engine(ModelSpec(D=D, K=c["K"], H=c["H"], M=c["M"], group_dims=c["dims"], gen_type=c["typ"]), device=dev)
  g = torch.Generator(device=dev).manual_seed(0)
  for n in ("enc_Wmu", "enc_Wsig", "gen_W2"):
    eng.p[n].copy_((torch.rand(eng.p[n].shape, generator=g, device=dev) * 2 - 1) / max(1, eng.p[n].shape[-1]) ** 0.5)
  eng.p["log_sigma"].fill_(float(np.log(1e-2)))
  eng.p["Ws"].copy_(torch.randn(eng.p["Ws"].shape, generator=g, device=dev))
  X = torch.rand(c["B"], D, generator=g, device=dev)
  eps = torch.randn(c["S"], c["B"], c["K"], generator=g, device=dev)
  ev = []

  def step():
    eng.forward_backward(X, eps, 1.0)
    eng.adam_step(1e-3)
    eng.sgd_prox_step(1e-4, 1e-4)

  def one_launch():
    eng.small_step(X, eps, 1.0, 1e-4, lr=1e-3, Ws_lr=1e-4)

  fused = eng.small_step_applies(c["B"], c["S"])
  dt_fused = None
  if fused:   # the one-launch step (csrc/small_step.hip), untraced: a pair of events costs as much as the launch
    for _ in range(5):
      one_launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
      one_launch()
    torch.cuda.synchronize()
    dt_fused = (time.perf_counter() - t0) / 200
  for _ in range(5):
    step()
  torch.cuda.synchronize()
  n = 50
  t0 = time.perf_counter()
  for _ in range(n):
    step()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / n
  eng.trace = ev
  for _ in range(10):
    step()
  torch.cuda.synchronize()
  eng.trace = None
  per = {}
  for nm, a, b in ev:
    per.setdefault(nm, []).append(a.elapsed_time(b))
  best = dt_fused if fused else dt
  print(json.dumps({"config": name, "ms_per_step": best * 1e3, "steps_per_s": 1 / best, "samples_per_s": c["B"] / best,
                    "path": "one launch (oivae_small_step_f32)" if fused else "entry points",
                    "ms_per_step_entry_points": dt * 1e3,
                    "entry_ms": {k: round(sum(v) / len(v), 4) for k, v in per.items()}}), flush=True)
