"""Structure of the gen_W2 gradient error in a dead group (Ws[g] = 0) at a long batch, default kernels vs float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oi_vae_amd import _lib
from tests._f64_reference import f64_step
from tests.test_gpu_scale import _engine
torch.set_printoptions(precision=2, linewidth=200, sci_mode=True)
G, K = int(os.environ.get("DIAG_G","16")), 64
DEAD = slice(100,108) if G>=128 else slice(4,8)
for B in (4096, 16384, 65536):
  for variant in ("dead only",):
    eng = _engine(G, 64, 64, 64, 64, seed=3)
    g = torch.Generator(device="cuda").manual_seed(17)
    X = torch.rand(B, G * 64, generator=g, device="cuda")
    eps = torch.randn(1, B, K, generator=g, device="cuda")
    eng.p["Ws"][DEAD] = 0.0
    if variant in ("dead + sigma spread", "all"):
      eng.p["log_sigma"].copy_(torch.log(10 ** (torch.rand(G * 64, generator=g, device="cuda") * 1.5 - 2.5)))
    if variant in ("dead + W scale", "all"):
      eng.p["gen_W2"].mul_((0.3 + 2.0 * torch.rand(G, 1, 1, generator=g, device="cuda")).expand(G, 64, 1).reshape(-1, 1))
      eng.p["gen_W1"].mul_(0.3 + 2.0 * torch.rand(G, 1, 1, generator=g, device="cuda"))
    sc, ref = f64_step(eng, X, eps, 0.6, rows_per_chunk=4096)
    line = []
    for tag, hooks in (("default", {}), ("g", dict(OIVAE_DEC="g")), ("bf16x3", dict(OIVAE_ENC="bf16", OIVAE_DEC="p")), ("fp32", dict(OIVAE_ENC="fp32", OIVAE_DEC="v4"))):
      _lib.set_test_hooks(OIVAE_ENC=hooks.get("OIVAE_ENC"), OIVAE_DEC=hooks.get("OIVAE_DEC"))
      eng._ws_key = None
      eng.forward_backward(X, eps, 0.6)
      torch.cuda.synchronize()
      r = ref["gen_W2"].reshape(G, 64, 64); q = eng.g["gen_W2"].double().reshape(G, 64, 64)
      e = (q - r).abs().flatten(1).amax(1) / r.abs().flatten(1).amax(1)
      line.append("%s live %.1e dead %.1e" % (tag, float(torch.cat([e[:DEAD.start], e[DEAD.stop:]]).max()), float(e[DEAD].max())))
      if False:
        w = DEAD.start + int(e[DEAD].argmax()); print("  worst dead group", w)
        d = (q[w] - r[w]) / r[w].abs().max()
        print("  dead group 4: err/scale by output dim d (max over m):", d.abs().amax(1)[:16])
        print("  by hidden unit m (max over d):", d.abs().amax(0)[:16])
        print("  signed err[d=0..3, m=0..7]:", d[:4, :8])
        print("  ref[d=0..3, m=0..7]/scale:", (r[w] / r[w].abs().max())[:4, :8])
        rb = ref["gen_b2"].reshape(G, 64); qb = eng.g["gen_b2"].double().reshape(G, 64)
        print("  gen_b2 rel err dead group:", float(((qb[w]-rb[w]).abs().max()/rb[w].abs().max())))
    _lib.set_test_hooks(OIVAE_ENC=None, OIVAE_DEC=None)
    print("B=%d %-22s %s" % (B, variant, " | ".join(line)))
