"""Recognise a model built with the reference's API and move it into an Engine.

The reference lets callers compose arbitrary torch modules; its scripts only
ever build three generator shapes and one encoder shape (SURVEY.md section 8a,
rows E and D).  `adopt` pattern-matches exactly those, reads the constants of the
opaque lambda heads by probing them in float64, and re-points every parameter
tensor at a view of the engine's flat device buffers.  Anything else is refused
with a clear error: the engine hard-codes its heads, it does not trace Python.
"""
import math

import torch

from . import _lib
from .engine import Engine, ModelSpec
from .utils import Lambda


def _as_list(net):
  if isinstance(net, torch.nn.Sequential):
    return list(net)
  return [net]


def _probe_affine_of_exp(mods):
  """mods: modules applied after the sigma head's Linear.  They must compute
  exp(a)*c + off.  Returns (c, off), probed in float64."""
  def f(a):
    x = torch.tensor([a], dtype=torch.float64)
    for m in mods:
      x = m(x)
    return float(x[0])
  f0, f1, fm = f(0.0), f(1.0), f(-1.0)
  c = (f1 - f0) / (math.e - 1.0)
  off = f0 - c
  if abs(fm - (c / math.e + off)) > 1e-9 * max(1.0, abs(fm)):
    raise _lib.OivaeError("encoder sigma head is not exp(a)*c + offset; oi_vae_amd supports only that head "
                          "(bars_data_oivae.py:81-86)")
  return c, off


def probe_encoder_head(inference_net, cache=None):
  """(c, offset) of the encoder's sigma head exp(a)*c + offset, read by probing the modules behind its Linear in
  float64 -- every step, because the scripts change c behind the model's back (a global captured by a lambda,
  bars_data_oivae.py:60,85,258-259).  `cache` (a dict owned by the caller): one probe per step instead of three -- the
  head is evaluated at 0 only, and while that value is the last step's, so are c and the offset."""
  mods = _as_list(inference_net.sigma_net)[1:]
  if cache is not None:
    x = torch.zeros(1, dtype=torch.float64)
    for m in mods:
      x = m(x)
    f0 = float(x[0])
    if cache.get("f0") == f0:
      return cache["val"]
    cache["f0"], cache["val"] = f0, _probe_affine_of_exp(mods)
    return cache["val"]
  return _probe_affine_of_exp(mods)


def _probe_gen_sigma(sigma_net, dg):
  if not (isinstance(sigma_net, Lambda) and len(sigma_net.extra_args) >= 1 and torch.is_tensor(sigma_net.extra_args[0])):
    raise _lib.OivaeError("generator sigma_net must be Lambda(f, extra_args=(log_sigma,)) (bars_data_oivae.py:98-101)")
  ls = sigma_net.extra_args[0]
  if ls.numel() != dg:
    raise _lib.OivaeError("log_sigma has %d entries for a group of %d outputs" % (ls.numel(), dg))
  x = torch.zeros(1, 1, dtype=torch.float64)
  f0 = sigma_net.func(x, torch.zeros(dg, dtype=torch.float64))
  f1 = sigma_net.func(x, torch.ones(dg, dtype=torch.float64))
  off = float(f0.reshape(-1)[0]) - 1.0
  if abs(float(f1.reshape(-1)[0]) - (math.e + off)) > 1e-9 or tuple(f0.shape) != (1, dg):
    raise _lib.OivaeError("generator sigma head is not exp(log_sigma) + offset broadcast over rows")
  return ls, off


def _gen_structure(gen):
  mods = _as_list(gen.mu_net)
  kinds = [type(m) for m in mods]
  L, T = torch.nn.Linear, torch.nn.Tanh
  if kinds == [L]:
    return "linear", None, mods[0]
  if kinds == [T, L]:
    return "tanh_linear", None, mods[1]
  if kinds == [L, T, L]:
    return "mlp", mods[0], mods[2]
  raise _lib.OivaeError("unsupported generator mu_net %s: expected Linear | Tanh,Linear | Linear,Tanh,Linear" % kinds)


def adopt(inference_net, generative_net, device=None) -> Engine:
  enc_mu = _as_list(inference_net.mu_net)
  enc_sig = _as_list(inference_net.sigma_net)
  if len(enc_mu) != 1 or not isinstance(enc_mu[0], torch.nn.Linear) or not isinstance(enc_sig[0], torch.nn.Linear):
    raise _lib.OivaeError("encoder must be mu_net=Linear(D,K), sigma_net=Sequential(Linear(D,K), exp, affine)")
  lin_mu, lin_sig = enc_mu[0], enc_sig[0]
  K, D = lin_mu.weight.shape
  if tuple(lin_sig.weight.shape) != (K, D) or lin_mu.bias is None or lin_sig.bias is None:
    raise _lib.OivaeError("encoder heads must both be Linear(%d,%d) with bias" % (D, K))
  probe_encoder_head(inference_net)

  gens = list(generative_net.group_generators)
  structs = [_gen_structure(g) for g in gens]
  typ = structs[0][0]
  if any(s[0] != typ for s in structs):
    raise _lib.OivaeError("all group generators must share one architecture")
  G, Kz, H = generative_net.Ws.shape
  if Kz != K or G != len(gens):
    raise _lib.OivaeError("Ws is %s but dim_z=%d and there are %d generators" % (tuple(generative_net.Ws.shape), K, len(gens)))
  dims = [s[2].out_features for s in structs]
  M = structs[0][1].out_features if typ == "mlp" else 0
  hin = M if typ == "mlp" else H
  offs = []
  for s in structs:
    if s[2].in_features != hin or s[2].bias is None:
      raise _lib.OivaeError("last generator layer must be Linear(%d, D_g) with bias" % hin)
    if typ == "mlp" and (s[1].in_features != H or s[1].out_features != M or s[1].bias is None):
      raise _lib.OivaeError("first generator layer must be Linear(%d, %d) with bias" % (H, M))
  log_sigmas = []
  for gen, dg in zip(gens, dims):
    ls, off = _probe_gen_sigma(gen.sigma_net, dg)
    log_sigmas.append(ls)
    offs.append(off)
  if max(offs) - min(offs) > 1e-12:
    raise _lib.OivaeError("generator sigma offsets differ between groups")
  if sum(dims) != D:
    raise _lib.OivaeError("group output dims sum to %d, encoder input is %d" % (sum(dims), D))

  spec = ModelSpec(D=D, K=K, H=H, group_dims=dims, gen_type=typ, M=M)
  if device is None:
    cands = [lin_mu.weight, generative_net.Ws]
    device = next((t.device for t in cands if t.is_cuda), None)
  eng = Engine(spec, device=device)
  eng.gen_sig_off = offs[0]

  pairs = []  # (user tensor, param view, grad view)

  def bind(t, name, g=None):
    pv = eng.p[name] if g is None else eng.group_slices(name, g)
    gv = eng.g[name] if g is None else eng.group_slices(name, g, grads=True)
    pv.copy_(t.detach().to(pv.device).reshape(pv.shape))
    t.data = pv.view(t.shape)
    gv = gv.view(t.shape)
    t.grad = gv     # bound ONCE: the engine overwrites its flat gradient buffer completely on every step
    pairs.append((t, gv))

  bind(lin_mu.weight, "enc_Wmu"); bind(lin_mu.bias, "enc_bmu")
  bind(lin_sig.weight, "enc_Wsig"); bind(lin_sig.bias, "enc_bsig")
  for g, s in enumerate(structs):
    if typ == "mlp":
      bind(s[1].weight, "gen_W1", g); bind(s[1].bias, "gen_b1", g)
    bind(s[2].weight, "gen_W2", g); bind(s[2].bias, "gen_b2", g)
    bind(log_sigmas[g], "log_sigma", g)
  bind(generative_net.Ws, "Ws")
  eng.bound = pairs
  generative_net._engine = eng
  return eng
