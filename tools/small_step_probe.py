#!/usr/bin/env python3
"""Phase times inside small_step_kernel (a scratch build with -DOIVAE_SMALL_STAMPS: tools/build_small_stamps.sh):
wall_clock64 (100 MHz) at the phase boundaries of every workgroup, read back from the workspace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oi_vae_amd import Engine, ModelSpec
MOCAP = [3, 3, 3, 3, 3, 3, 3, 2, 3, 1, 1, 2, 1, 2, 2, 3, 1, 1, 2, 1, 2, 3, 1, 2, 1, 3, 1, 2, 1]
for name, c in {"bars": dict(dims=[8] * 8, K=16, H=1, typ="linear", B=64, S=1),
                "mocap": dict(dims=MOCAP, K=8, H=8, typ="tanh_linear", B=64, S=10)}.items():
  D = sum(c["dims"]); dev = torch.device("cuda", 0)
  eng = Engine(ModelSpec(D=D, K=c["K"], H=c["H"], M=0, group_dims=c["dims"], gen_type=c["typ"]), device=dev)
  g = torch.Generator(device=dev).manual_seed(0)
  for n in ("enc_Wmu", "enc_Wsig", "gen_W2"):
    eng.p[n].copy_((torch.rand(eng.p[n].shape, generator=g, device=dev) * 2 - 1) / max(1, eng.p[n].shape[-1]) ** 0.5)
  eng.p["log_sigma"].fill_(float(np.log(1e-2)))
  eng.p["Ws"].copy_(torch.randn(eng.p["Ws"].shape, generator=g, device=dev))
  X = torch.rand(c["B"], D, generator=g, device=dev); eps = torch.randn(c["S"], c["B"], c["K"], generator=g, device=dev)
  for _ in range(3):
    eng.small_step(X, eps, 1.0, 1e-4, lr=1e-3, Ws_lr=1e-4)
  torch.cuda.synchronize()
  ws = eng._ws.cpu().numpy()
  off = eng._ws_bytes - 4352 + 64     # the one-launch step's region is the workspace's last; its stamps follow the ticket counter
  st = np.frombuffer(ws[off:off + 8 * 16 * 32].tobytes(), dtype=np.uint64).reshape(-1, 16)
  t0 = min(int(r[0]) for r in st if r[0])
  for b, r in enumerate(st):
    if r[0]:
      print(name, "workgroup", b, "phase boundaries (us from the first start):", [round((int(x) - t0) / 100.0, 1) for x in r[:10] if x], '(last: end of the first round\'s item phase)')
