"""Per-group error of every gradient tensor at the full scale-up shape against the float64 reference
(tests/_f64_reference.py), under the default kernels and the cross-check families: tells a kernel's inaccuracy from
an ill-conditioned state (a sum over 65536 rows that cancels)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oi_vae_amd import _lib
from tests._f64_reference import f64_step
from tests.test_gpu_scale import _engine

B, K = int(os.environ.get("DIAG_B", "65536")), 64
G = int(os.environ.get("DIAG_G", "1024"))
eng = _engine(G, 64, 64, 64, 64, seed=3)
g = torch.Generator(device="cuda").manual_seed(17)
X = torch.rand(B, G * 64, generator=g, device="cuda")
eps = torch.randn(1, B, K, generator=g, device="cuda")
dead_rows = torch.rand(G, 64, generator=g, device="cuda") < 0.33
eng.p["Ws"][dead_rows] = 0.0
eng.p["Ws"][100:108] = 0.0
eng.p["log_sigma"].copy_(torch.log(10 ** (torch.rand(G * 64, generator=g, device="cuda") * 1.5 - 2.5)))
eng.p["gen_W2"].mul_((0.3 + 2.0 * torch.rand(G, 1, 1, generator=g, device="cuda")).expand(G, 64, 1).reshape(-1, 1))
eng.p["gen_W1"].mul_(0.3 + 2.0 * torch.rand(G, 1, 1, generator=g, device="cuda"))
sc, ref = f64_step(eng, X, eps, 0.6, rows_per_chunk=2048)
for tag, hooks in (("default", {}), ("bf16x3", dict(OIVAE_ENC="bf16", OIVAE_DEC="p")), ("fp32", dict(OIVAE_ENC="fp32", OIVAE_DEC="v4"))):
  _lib.set_test_hooks(OIVAE_ENC=hooks.get("OIVAE_ENC"), OIVAE_DEC=hooks.get("OIVAE_DEC"))
  eng._ws_key = None
  out = eng.forward_backward(X, eps, 0.6)
  torch.cuda.synchronize()
  rep = {}
  for name, r in ref.items():
    gq = eng.g[name].double()
    if name.startswith("gen_") or name == "log_sigma":
      d = (gq - r).abs().reshape(G, -1).amax(1); s = r.abs().reshape(G, -1).amax(1)
      e = d / s.clamp(min=1e-300)
      w = int(e.argmax())
      rep[name] = dict(worst=float(e.max()), group=w, scale_of_worst=float(s[w]), median_scale=float(s.median()),
                       n_over_2e5=int((e > 2e-5).sum()), whole=float(d.max() / s.max()))
    else:
      rep[name] = dict(worst=float((gq - r).abs().max() / r.abs().max()))
  print(tag, json.dumps(rep))
_lib.set_test_hooks(OIVAE_ENC=None, OIVAE_DEC=None)
