"""Seeded random shapes through the default kernel choice and through the shape-generic kernels (OIVAE_FORCE_GENERIC):
whatever path a shape is served by -- fp16-term or fp32 matrix decoders, padded / paired launches of narrower two-layer shapes, dec_single, the flat
many-small-groups path, h2 / fp32 / generic encoder GEMMs, ragged groups, batches that are no multiple of anything --
one step must give the same scalars (1e-5) and gradients (2e-5 of each tensor's scale), and the optimizers + proximal
step the same zero set.  The shapes are drawn, not chosen: the point is the combinations nobody wrote a test for."""
import os

import numpy as np
import pytest
import torch

from oi_vae_amd import _lib

pytestmark = pytest.mark.gpu


def _draw(seed):
  r = np.random.default_rng(seed)
  kind = r.choice(["linear", "tanh_linear", "mlp", "mlp_mix", "small_groups"])
  S = int(r.choice([1, 1, 2, 3]))
  B = int(r.choice([1, 7, 33, 64, 100, 257]))
  if kind == "mlp_mix":      # the matrix decoders' shapes: every dimension 32 or 64 ...
    K, H, M = (int(r.choice([32, 64])) for _ in range(3))
    dg = int(r.choice([32, 64]))
    G = int(r.integers(1, 6))
    if r.random() < 0.5:     # ... or any width up to 64 with enough work per group for the padded launch (kPadMinWork)
      for _ in range(100):
        K, H, M, dg = int(r.integers(1, 65)), int(r.integers(1, 65)), int(r.integers(1, 65)), 4 * int(r.integers(1, 17))
        if K * H + H * M + M * dg >= 1536:
          break
    return dict(gen_type="mlp", K=K, H=H, M=M, dims=[dg] * G, B=B, S=S)
  if kind == "small_groups":   # towards the flat path: K a multiple of 128, G H a multiple of 128, uniform tiny groups
    H = int(r.choice([4, 8]))
    return dict(gen_type="linear", K=128, H=H, M=0, dims=[int(r.choice([2, 4]))] * (128 // H * int(r.integers(1, 3))),
                B=int(r.choice([32, 96])), S=1)
  K = int(r.choice([1, 2, 3, 8, 10, 16, 31, 32, 40]))
  H = int(r.choice([1, 2, 4, 8, 16, 5]))
  G = int(r.integers(1, 9))
  dims = [int(x) for x in r.integers(1, 40, G)]
  if r.random() < 0.5:       # rows of X 16-byte aligned: the matrix-core encoders apply
    dims[-1] += (-sum(dims)) % 4
  M = int(r.choice([8, 24, 33])) if kind == "mlp" else 0
  return dict(gen_type=kind if kind != "mlp" else "mlp", K=K, H=H, M=M, dims=dims, B=B, S=S)


def _run(cfg, seed, generic):
  from oi_vae_amd import Engine, ModelSpec
  _lib.set_test_hooks(OIVAE_FORCE_GENERIC="1" if generic else None)
  try:
    D = sum(cfg["dims"])
    eng = Engine(ModelSpec(D=D, K=cfg["K"], H=cfg["H"], M=cfg["M"], group_dims=cfg["dims"], gen_type=cfg["gen_type"]),
                 device="cuda:0")
    g = torch.Generator(device="cuda").manual_seed(1000 + seed)
    for n, t in eng.p.items():
      if n == "log_sigma":
        t.copy_(torch.log(10 ** (torch.rand(t.shape, generator=g, device="cuda") * 1.5 - 2.0)))
      elif n == "Ws":
        t.copy_(torch.randn(t.shape, generator=g, device="cuda"))
      else:
        t.copy_((torch.rand(t.shape, generator=g, device="cuda") * 2 - 1) / max(1, t.shape[-1]) ** 0.5)
    if seed % 3 == 0 and eng.p["Ws"].size(0) > 1:
      eng.p["Ws"][0] = 0.0      # a dead group
    X = torch.rand(cfg["B"], D, generator=g, device="cuda")
    eps = torch.randn(cfg["S"], cfg["B"], cfg["K"], generator=g, device="cuda")
    out = eng.forward_backward(X, eps, 0.7)
    grads, sc = eng.flat_grads.clone(), out["scalars"].clone()
    t = float(torch.sqrt((eng.p["Ws"] ** 2).sum(-1)).median())
    eng.adam_step(1e-3)
    eng.sgd_prox_step(1e-4, t)
    torch.cuda.synchronize()
    eng.check_errors()
    return eng, grads, sc, eng.flat_params.clone(), eng.zero_mask.clone()
  finally:
    _lib.set_test_hooks(OIVAE_FORCE_GENERIC=None)


@pytest.mark.parametrize("seed", range(int(os.environ.get("OIVAE_FUZZ_SEEDS", "48"))))   # (a one-off run of 800 seeds: all green)
def test_random_shape_default_path_matches_generic_kernels(seed):
  cfg = _draw(seed)
  eng, ga, sa, pa, za = _run(cfg, seed, generic=False)
  _, gb, sb, pb, zb = _run(cfg, seed, generic=True)
  assert ((sa[:6] - sb[:6]).abs() / sb[:6].abs().clamp(min=1)).max().item() < 1e-5, (cfg, sa[:6].tolist(), sb[:6].tolist())
  for name in eng.seg_off:
    o, n = eng.seg_off[name], eng.seg_numel[name]
    if n == 0:
      continue
    a, b = ga[o:o + n], gb[o:o + n]
    assert torch.isfinite(a).all(), (cfg, name)
    err = ((a - b).abs().max() / b.abs().max().clamp(min=1e-30)).item()
    assert err < 2e-5, (cfg, name, err)
    # post-step weights.  Adam's first step is lr * g / (|g| + 1e-8), i.e. about lr * sign(g): an element whose
    # gradient is rounding noise (below 1e-3 of the tensor's scale here; seed 534 has one of 1.5e-3 against -3.0e-4 in
    # a tensor of scale 8.7e4) may legitimately move by +lr on one path and -lr on the other; everywhere else 1e-5
    pa_, pb_ = pa[o:o + n], pb[o:o + n]
    solid = b.abs() >= 1e-3 * b.abs().max()
    pscale = pb_.abs().max().clamp(min=1e-30)
    if name != "Ws":
      assert ((pa_ - pb_).abs() * solid).max().item() < 1e-5 * pscale.item(), (cfg, name, "post-step")
      assert (pa_ - pb_).abs().max().item() <= 2.0 * 1e-3 + 1e-5 * pscale.item(), (cfg, name, "post-step, noise-level gradients")
    else:   # SGD is linear in the gradient; rows at the threshold are covered by the zero-set check below
      same = (za == zb).repeat_interleave(eng.spec.H).view(-1)
      assert ((pa_ - pb_).abs() * same).max().item() < 1e-5 * pscale.item(), (cfg, name, "post-step")
  # rows within rounding of the threshold may fall either way when the gradients differ in the last bits: at most one
  assert int((za != zb).sum()) <= 1, (cfg, int((za != zb).sum()))


@pytest.mark.parametrize("seed", range(int(os.environ.get("OIVAE_FUZZ_ORACLE_SEEDS", "24"))))
def test_random_shape_default_path_matches_the_cpu_oracle(seed):
  """The same drawn shapes against the CPU oracle (oracle/oivae_oracle.py, pinned to the reference by the golden
  fixtures) run live: scalars 1e-5, gradients 2e-5 of each reference tensor's scale, on the oracle's own initial
  parameters with a dead group now and then."""
  from oi_vae_amd import Engine, ModelSpec
  from oracle import oivae_oracle as O
  from tests._fixtures import flat_items, rel_err
  cfg = _draw(seed + 1000)
  D = sum(cfg["dims"])
  spec = dict(D=D, K=cfg["K"], H=cfg["H"], M=cfg["M"], gen_type=cfg["gen_type"], group_dims=cfg["dims"])
  params = O.init_params(spec, seed=seed)
  if seed % 3 == 0 and len(cfg["dims"]) > 1:
    params["Ws"][0] = 0.0
  rng = np.random.default_rng(seed)
  X = rng.random((cfg["B"], D), dtype=np.float32)
  eps = rng.standard_normal((cfg["S"], cfg["B"], cfg["K"])).astype(np.float32)
  tr = O.OracleTrainer(spec, params, lr=1e-3, Ws_lr=1e-4, lam=1.0)
  info = tr.step(X, eps, c=0.7, prox_step_size=1e-4)
  eng = Engine(ModelSpec(D=D, K=cfg["K"], H=cfg["H"], M=cfg["M"], group_dims=cfg["dims"], gen_type=cfg["gen_type"]),
               device="cuda:0")
  eng.load_params(params)
  out = eng.forward_backward(X, eps, 0.7)
  torch.cuda.synchronize()
  eng.check_errors()
  sc = out["scalars"].cpu().numpy()
  for j, k in enumerate(("z_kl", "loglik_term", "logprob_theta", "logprob_W", "loss", "elbo")):
    assert abs(sc[j] - info[k]) <= 1e-5 * max(abs(info[k]), 1.0), (cfg, k, sc[j], info[k])
  got = dict(flat_items(eng.export_grads()))
  # a one-row batch has nothing to average a cancelling K-term dot product's fp32 rounding over: seed 209 (B = 1,
  # K = 3, H = 1) has a group whose h = z . Ws cancels to 2.3e-5 of its terms, and every gradient that is a multiple
  # of h differs from the oracle's -- equally fp32 -- value by that much; from a handful of rows on, 2e-5
  bar = 2e-5 if cfg["B"] * cfg["S"] > 4 else 1e-4
  for k, b in flat_items(info["grads"]):
    assert rel_err(got[k], b) < bar, (cfg, k, rel_err(got[k], b))


def _draw_small(seed):
  r = np.random.default_rng(5000 + seed)
  H = int(r.choice([1, 2, 4, 8]))
  K = int(r.integers(1, 33))
  G = int(r.integers(1, 30))
  dims = [int(x) for x in r.integers(1, 17, G)]
  while sum(dims) > 256:
    dims.pop()
  S = int(r.choice([1, 1, 2, 5, 10]))
  B = int(r.choice([1, 5, 37, 64, 100, 200]))
  return dict(gen_type=str(r.choice(["linear", "tanh_linear"])), K=K, H=H, M=0, dims=dims, B=B, S=S)


@pytest.mark.parametrize("seed", range(int(os.environ.get("OIVAE_FUZZ_SMALL_SEEDS", "24"))))
def test_random_small_model_one_launch_step_matches_the_entry_points(seed):
  """Drawn small single-layer models (the shapes oivae_small_step_f32 accepts; ragged groups, odd batches, several
  samples): three steps of the one-launch kernel -- optimizers and proximal step inside -- against three steps of the
  entry points on a twin engine: scalars 1e-5, gradients 2e-5, the same zero set, weights 1e-5 where Adam's first
  steps are not deciding on noise."""
  from oi_vae_amd import Engine, ModelSpec
  cfg = _draw_small(seed)
  D = sum(cfg["dims"])
  twins = []
  for _ in range(2):
    eng = Engine(ModelSpec(D=D, K=cfg["K"], H=cfg["H"], M=0, group_dims=cfg["dims"], gen_type=cfg["gen_type"]), device="cuda:0")
    g = torch.Generator(device="cuda").manual_seed(77 + seed)
    for n, t in eng.p.items():
      if n == "log_sigma":
        t.fill_(float(np.log(1e-1)))
      elif n == "Ws":
        t.copy_(torch.randn(t.shape, generator=g, device="cuda"))
      elif t.numel():
        t.copy_((torch.rand(t.shape, generator=g, device="cuda") * 2 - 1) / max(1, t.shape[-1]) ** 0.5)
    twins.append(eng)
  a, b = twins
  if not a.small_step_applies(cfg["B"], cfg["S"]):
    pytest.skip("outside the one-launch step: %r" % (cfg,))
  g = torch.Generator(device="cuda").manual_seed(9 + seed)
  t_prox = float(torch.sqrt((a.p["Ws"] ** 2).sum(-1)).median()) * 0.5
  for step in range(3):
    X = torch.rand(cfg["B"], D, generator=g, device="cuda")
    eps = torch.randn(cfg["S"], cfg["B"], cfg["K"], generator=g, device="cuda")
    oa = a.small_step(X, eps, 0.7, t_prox, lr=1e-3, Ws_lr=1e-3)
    ob = b.forward_backward(X, eps, 0.7)
    sa, sb = oa["scalars"].clone(), ob["scalars"].clone()
    ga, gb = a.flat_grads.clone(), b.flat_grads.clone()
    b.adam_step(1e-3)
    b.sgd_prox_step(1e-3, t_prox)
    torch.cuda.synchronize()
    assert ((sa[:6] - sb[:6]).abs() / sb[:6].abs().clamp(min=1)).max().item() < 1e-5, (cfg, step)
    for name in a.seg_off:
      o, n = a.seg_off[name], a.seg_numel[name]
      if n == 0:
        continue
      x, y = ga[o:o + n], gb[o:o + n]
      # single rows: dz of ONE row and a latent or two is a sum over the groups that can cancel to a fraction of a
      # percent of its terms (seed 322, third step: encoder gradients of scale 2.9 where the steps before had 990 and
      # 650) -- both paths then carry the same fp32 rounding of the terms, 2e-4 of what is left
      # (and a few rows more do not average it away yet: seed 1713 -- 5 rows x 5 samples, ONE latent, 29 groups -- has
      # enc_bmu = -7.6900 against -7.6902, 2.3e-5, on its second step; from 64 decoder rows on, 2e-5)
      rows = cfg["B"] * cfg["S"]
      bar = 2e-5 if rows >= 64 else (1e-4 if rows > 4 else 1e-3)
      assert ((x - y).abs().max() / y.abs().max().clamp(min=1e-30)).item() < bar, (cfg, step, name)
    assert int((a.zero_mask != b.zero_mask).sum()) <= 1, (cfg, step)
    # keep the twins on one trajectory: what is compared is a step, not the drift of three
    b.flat_params.copy_(a.flat_params)
    b.adam_m.copy_(a.adam_m); b.adam_v.copy_(a.adam_v)
