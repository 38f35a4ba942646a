"""GPU parity tests: the HIP engine, called through the C ABI, against the
oracle and the golden vectors recorded from the real reference.

Tolerances (BASELINE.json north_star): per-step ELBO terms and weights within
1e-5 relative (relative to the tensor's largest magnitude for tensors); the zero
set of Ws rows after the proximal step bit-exact; the proximal step itself
bit-exact given the same input bits (teacher-forced)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from tests._fixtures import GOLDEN, STEP_CONFIGS, elementwise_within, flat_items, load, params_at, rel_err

pytestmark = pytest.mark.gpu

SCALARS = ("z_kl", "loglik_term", "logprob_theta", "logprob_W", "loss", "elbo")


def _engine(spec):
  from oi_vae_amd import Engine, ModelSpec
  ms = ModelSpec(D=spec["D"], K=spec["K"], H=spec["H"], group_dims=spec["group_dims"],
                 gen_type=spec["gen_type"], M=spec.get("M", 0))
  return Engine(ms, device="cuda:0")


def _prox_gpu(W, t, with_grad=None, lr=0.0):
  from oi_vae_amd import _lib
  L = _lib.init_device(0)
  G, K, H = W.shape
  w = torch.from_numpy(W.copy()).cuda()
  mask = torch.zeros(G, K, dtype=torch.uint8, device="cuda")
  norms = torch.zeros(G, K, dtype=torch.float32, device="cuda")
  st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  if with_grad is None:
    _lib.check(L.oivae_prox_f32(w.data_ptr(), G * K, H, float(t), mask.data_ptr(), norms.data_ptr(), st), "prox")
  else:
    g = torch.from_numpy(with_grad.copy()).cuda()
    _lib.check(L.oivae_sgd_prox_f32(w.data_ptr(), g.data_ptr(), G * K, H, float(lr), float(t), 1,
                                    mask.data_ptr(), norms.data_ptr(), st), "sgd_prox")
  torch.cuda.synchronize()
  return w.cpu().numpy(), mask.cpu().numpy(), norms.cpu().numpy()


def test_prox_golden_vectors_bit_exact():
  z = np.load(os.path.join(GOLDEN, "prox_vectors.npz"))
  for H in z["Hs"]:
    W = z["H%d/W" % H]
    for key, t in (("out", float(z["H%d/t" % H])), ("out_small_t", 1e-4)):
      got, mask, _ = _prox_gpu(W, t)
      want = z["H%d/%s" % (H, key)]
      assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "H=%d %s" % (H, key)
      assert np.array_equal(mask, np.all(want == 0, axis=-1).astype(np.uint8)), "H=%d %s mask" % (H, key)


def test_wide_prox_golden_vectors_bit_exact():
  """H >= 512 (prox_rows_wide: 32 lanes per row, one cascading accumulator chain each) against the vectors recorded
  from the reference, and with the fused SGD update against the C oracle."""
  from oracle import prox_oracle
  z = np.load(os.path.join(GOLDEN, "prox_vectors_wide.npz"))
  rng = np.random.default_rng(3)
  for H in z["Hs"]:
    W = z["H%d/W" % H]
    keys = [("out", float(z["H%d/t" % H]))] + ([("out_small_t", 1e-4)] if "H%d/out_small_t" % H in z else [])
    for key, t in keys:
      got, mask, _ = _prox_gpu(W, t)
      want = z["H%d/%s" % (H, key)]
      assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "H=%d %s" % (H, key)
      assert np.array_equal(mask, np.all(want == 0, axis=-1).astype(np.uint8)), "H=%d %s mask" % (H, key)
    g = (rng.standard_normal(W.shape) * 20).astype(np.float32)
    lr, t = 0.0078125, float(z["H%d/t" % H])
    pre = (W.astype(np.float64) - lr * g.astype(np.float64)).astype(np.float32)
    want, wmask, wnorms = prox_oracle.proximal_step(pre, t)
    got, mask, norms = _prox_gpu(W, t, with_grad=g, lr=lr)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "H=%d sgd" % H
    assert np.array_equal(mask, wmask) and np.array_equal(norms.view(np.uint32), wnorms.view(np.uint32))
  # many rows of a wide H: every 32-lane group of the launch, partial last workgroup included
  W = rng.standard_normal((37, 11, 1000)).astype(np.float32) * rng.random((37, 11, 1)).astype(np.float32)
  t = float(np.median(np.sqrt((W.astype(np.float64) ** 2).sum(-1))))
  want, wmask, wnorms = prox_oracle.proximal_step(W, t)
  got, mask, norms = _prox_gpu(W, t)
  assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
  assert np.array_equal(mask, wmask) and np.array_equal(norms.view(np.uint32), wnorms.view(np.uint32))
  assert 0.3 < mask.mean() < 0.7


def test_prox_kernels_sweep_of_widths_vs_oracle_bit_exact():
  """Every kernel path of the row form -- scalar (any H), two threads per row (H = 8, 16, 32, 64, 128), 32 threads per row
  (H >= 512) -- on a sweep of widths against the C oracle, with and without the fused SGD update."""
  from oracle import prox_oracle
  rng = np.random.default_rng(11)
  for H in list(range(1, 70)) + list(range(70, 1100, 41)) + [128, 256, 511, 512, 513, 543, 544, 2048, 4096, 4104, 8200]:
    W = (rng.standard_normal((3, 5, H)) * rng.uniform(0.2, 2.0)).astype(np.float32)
    W[1, 2] = 0
    t = float(np.median(np.sqrt((W.astype(np.float64) ** 2).sum(-1))))
    want, wmask, wnorms = prox_oracle.proximal_step(W, t)
    got, mask, norms = _prox_gpu(W, t)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), H
    assert np.array_equal(mask, wmask) and np.array_equal(norms.view(np.uint32), wnorms.view(np.uint32)), H
    if H % 3 == 0:
      g = (rng.standard_normal(W.shape) * 8).astype(np.float32)
      lr = 0.03125
      pre = (W.astype(np.float64) - lr * g.astype(np.float64)).astype(np.float32)
      want, wmask, _ = prox_oracle.proximal_step(pre, t)
      got, mask, _ = _prox_gpu(W, t, with_grad=g, lr=lr)
      assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), ("sgd", H)
      assert np.array_equal(mask, wmask), ("sgd", H)


def test_prox_random_vs_oracle_bit_exact():
  from oracle import prox_oracle
  rng = np.random.default_rng(7)
  for (G, K, H) in [(8192, 256, 8), (1024, 64, 64), (333, 7, 1), (40, 10, 5), (64, 33, 16), (17, 9, 100), (50, 20, 128),
                    (13, 7, 300), (5, 9, 511), (6, 5, 519)]:
    W = rng.standard_normal((G, K, H)).astype(np.float32)
    t = float(np.median(np.sqrt((W.astype(np.float64) ** 2).sum(-1))))
    got, mask, norms = _prox_gpu(W, t)
    want, wmask, wnorms = prox_oracle.proximal_step(W, t)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (G, K, H)
    assert np.array_equal(mask, wmask)
    assert np.array_equal(norms.view(np.uint32), wnorms.view(np.uint32))
    assert 0.3 < mask.mean() < 0.7


def test_sgd_prox_fused_matches_sgd_then_prox():
  from oracle import prox_oracle
  rng = np.random.default_rng(8)
  for (G, K, H) in [(64, 32, 8), (16, 64, 64), (9, 5, 3)]:
    W = rng.standard_normal((G, K, H)).astype(np.float32)
    g = (rng.standard_normal((G, K, H)) * 50).astype(np.float32)
    lr = 1e-2
    # torch.optim.SGD on CPU rounds w + (-lr)*g once (fma), see include/oivae_hip.h
    pre = (W.astype(np.float64) + np.float64(np.float32(-lr)) * g.astype(np.float64)).astype(np.float32)
    want, wmask, _ = prox_oracle.proximal_step(pre, 1.5)
    got, mask, _ = _prox_gpu(W, 1.5, with_grad=g, lr=lr)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (G, K, H)
    assert np.array_equal(mask, wmask)


@pytest.mark.parametrize("name", STEP_CONFIGS)
def test_step_matches_reference_golden(name):
  z, spec = load(name)
  eng = _engine(spec)
  eng.load_params(params_at(z, spec, "init/"))
  for i in range(spec["n_steps"]):
    pre = "step%d/" % i
    X, eps = z[pre + "X"], z[pre + "eps"]
    c, t = float(z[pre + "c"]), float(z[pre + "t"])
    out = eng.forward_backward(X, eps, c, sigma_theta=spec["sigma_theta"], lam=spec["lam"])
    torch.cuda.synchronize()
    sc = out["scalars"].cpu().numpy()
    for j, k in enumerate(SCALARS):
      want = float(z[pre + k])
      assert abs(sc[j] - want) <= 1e-5 * max(abs(want), 1.0), (name, i, k, sc[j], want)
    assert rel_err(out["mu_q"].cpu().numpy(), z[pre + "mu_q"]) < 1e-5
    assert rel_err(out["sig_q"].cpu().numpy(), z[pre + "sig_q"]) < 1e-5
    got_g = dict(flat_items(eng.export_grads()))
    for k, b in flat_items(params_at(z, spec, pre + "grad/")):
      assert rel_err(got_g[k], b) < 2e-5, (name, i, "grad", k, rel_err(got_g[k], b))
    # optimizers + prox, engine-native
    eng.adam_step(spec["lr"])
    eng.sgd_prox_step(spec["Ws_lr"], t, do_prox=spec["lam"] > 0)
    torch.cuda.synchronize()
    got_p = dict(flat_items(eng.export_params()))
    for k, b in flat_items(params_at(z, spec, pre + "post/")):
      assert rel_err(got_p[k], b) < 1e-5, (name, i, "param", k, rel_err(got_p[k], b))
    assert np.array_equal(eng.zero_mask.cpu().numpy(), z[pre + "zero_mask"]), (name, i, "zero set")
    # teacher-forced prox: the reference's own pre-prox bits in, its bits out
    got, mask, _ = _prox_gpu(z[pre + "pre_prox_Ws"], t)
    assert np.array_equal(got.view(np.uint32), z[pre + "post/Ws"].view(np.uint32)), (name, i)
    assert np.array_equal(mask, z[pre + "zero_mask"])


def test_sparse_trained_run_matches_reference_golden():
  """c4_trained64 (tests/golden/make_golden.py): 30 free-running steps of the engine at the all-64 shape -- the
  fp16-term encoder kernels and dec_x3_64g, five groups (an odd count: the last decoder workgroup holds one group) --
  against the reference's own run: a group that is dead from step 1, a partner group almost dead, rows dying at
  every step (150 of 320 zero at the peak) and reviving, generator weights and log_sigma driven far from their
  initial values by lr = 2e-2, the std head switched 0.1 -> 1 at step 10, a ragged S = 2 batch at the end.
  Scalars 1e-5 relative and the zero set exact at EVERY step, free-running; the teacher-forced proximal step bit-exact
  and gradients / post-step parameters at the steps the fixture holds them for: step 0 at the usual 2e-5 / 1e-5 of
  scale, steps 11, 12, 28 and 29 at 5e-5 / 3e-5 -- by then the two runs are on their own trajectories.  The PER-STEP bar
  at those steps (2e-5 / 1e-5 from the reference's own state) is test_teacher_forced_steps_of_the_trained_run below,
  and whose drift this is -- the kernels' or any fp32 trajectory's -- is test_free_running_drift_... ."""
  z, spec = load("c4_trained64")
  eng = _engine(spec)
  eng.load_params(params_at(z, spec, "init/"))
  for i in range(spec["n_steps"]):
    pre = "step%d/" % i
    X, eps = z[pre + "X"], z[pre + "eps"]
    c, t = float(z[pre + "c"]), float(z[pre + "t"])
    out = eng.forward_backward(X, eps, c, sigma_theta=spec["sigma_theta"], lam=spec["lam"])
    torch.cuda.synchronize()
    sc = out["scalars"].cpu().numpy()
    assert np.isfinite(sc[:6]).all(), (i, sc)
    for j, k in enumerate(SCALARS):
      want = float(z[pre + k])
      assert abs(sc[j] - want) <= 1e-5 * max(abs(want), 1.0), (i, k, sc[j], want)
    full = (pre + "post/Ws") in z.files
    gtol, ptol = (2e-5, 1e-5) if i == 0 else (5e-5, 3e-5)   # free-running; the per-step bar is the teacher-forced test
    if full:
      got_g = dict(flat_items(eng.export_grads()))
      for k, b in flat_items(params_at(z, spec, pre + "grad/")):
        assert rel_err(got_g[k], b) < gtol, (i, "grad", k, rel_err(got_g[k], b))
    eng.adam_step(spec["lr"])
    eng.sgd_prox_step(spec["Ws_lr"], t, do_prox=spec["lam"] > 0)
    torch.cuda.synchronize()
    assert np.array_equal(eng.zero_mask.cpu().numpy(), z[pre + "zero_mask"]), (i, "zero set")
    if full:
      got_p = dict(flat_items(eng.export_params()))
      for k, b in flat_items(params_at(z, spec, pre + "post/")):
        assert rel_err(got_p[k], b) < ptol, (i, "param", k, rel_err(got_p[k], b))
      got, mask, _ = _prox_gpu(z[pre + "pre_prox_Ws"], t)
      assert np.array_equal(got.view(np.uint32), z[pre + "post/Ws"].view(np.uint32)), i
      assert np.array_equal(mask, z[pre + "zero_mask"])


def _trained_run(hooks=None, upto=30):
  """The free-running c4_trained64 run under a kernel family; returns {step: (worst gradient error, worst parameter
  error)} over the steps the fixture holds in full, errors relative to each reference tensor's scale."""
  from oi_vae_amd import _lib
  z, spec = load("c4_trained64")
  hooks = hooks or {}
  drift = {}
  try:
    _lib.set_test_hooks(OIVAE_ENC=hooks.get("OIVAE_ENC"), OIVAE_DEC=hooks.get("OIVAE_DEC"))
    eng = _engine(spec)
    eng.load_params(params_at(z, spec, "init/"))
    for i in range(upto):
      pre = "step%d/" % i
      eng.forward_backward(z[pre + "X"], z[pre + "eps"], float(z[pre + "c"]), sigma_theta=spec["sigma_theta"], lam=spec["lam"])
      full = (pre + "post/Ws") in z.files
      if full:
        got_g = dict(flat_items(eng.export_grads()))
        ge = max(rel_err(got_g[k], b) for k, b in flat_items(params_at(z, spec, pre + "grad/")))
      eng.adam_step(spec["lr"])
      eng.sgd_prox_step(spec["Ws_lr"], float(z[pre + "t"]), do_prox=spec["lam"] > 0)
      torch.cuda.synchronize()
      assert np.array_equal(eng.zero_mask.cpu().numpy(), z[pre + "zero_mask"]), (hooks, i, "zero set")
      if full:
        got_p = dict(flat_items(eng.export_params()))
        pe = max(rel_err(got_p[k], b) for k, b in flat_items(params_at(z, spec, pre + "post/")))
        drift[i] = (ge, pe)
  finally:
    _lib.set_test_hooks(OIVAE_ENC=None, OIVAE_DEC=None)
  return drift


def test_free_running_drift_of_the_default_kernels_against_full_width_kernels():
  """Whose drift is it?  The same 30 free-running steps under the default kernels (operands as two scaled fp16
  terms), under the exact-operand kernels (three bf16 terms: OIVAE_ENC=bf16 OIVAE_DEC=p) and under the fp32 matrix
  instructions (OIVAE_ENC=fp32 OIVAE_DEC=v4).  All three must keep the reference's zero set at every step, and at
  the late steps the default kernels may not be further from the reference than 1.5 x the fp32-instruction kernels
  are (floor: the oracle's own bar, 1e-5 of scale) -- a free-running fp32 trajectory in another summation order
  drifts this much whatever it is computed with."""
  runs = {"fp16x2 (default)": _trained_run(),
          "bf16x3": _trained_run({"OIVAE_ENC": "bf16", "OIVAE_DEC": "p"}),
          "fp32 mfma": _trained_run({"OIVAE_ENC": "fp32", "OIVAE_DEC": "v4"})}
  print()
  for name, d in runs.items():
    print("free-running drift vs reference, %-17s" % name,
          "  ".join("step %d: grads %.1e weights %.1e" % (i, g, p) for i, (g, p) in sorted(d.items())))
  dflt, f32 = runs["fp16x2 (default)"], runs["fp32 mfma"]
  late = (11, 12, 28, 29)
  for i in late:   # the weights: step by step
    assert dflt[i][1] <= 1.5 * max(f32[i][1], 1e-5), ("weights", i, dflt[i][1], f32[i][1])
  # the gradients of a single late step scatter between the families (measured, grads at steps 12 / 28 / 29: default
  # 2.0e-5 / 3.1e-5 / 2.7e-5, bf16x3 2.0e-5 / 1.9e-5 / 9.3e-6, fp32 instructions 1.3e-5 / 3.2e-5 / 1.2e-5 -- each is worst
  # somewhere): the bar is on the worst of the late steps
  assert max(dflt[i][0] for i in late) <= 1.5 * max(max(f32[i][0] for i in late), 1e-5), (dflt, f32)


def _teacher_forced(i, mode):
  """One step from the reference's own state after step i - 1; returns the figures after asserting the per-step bars."""
  from oi_vae_amd import _lib
  z, spec = load("c4_trained64")
  prev, pre = "step%d/" % (i - 1), "step%d/" % i
  try:
    if mode == "fp32":
      _lib.set_test_hooks(OIVAE_ENC="fp32", OIVAE_DEC="v4")
    eng = _engine(spec)
    eng.load_params(params_at(z, spec, prev + "post/"))
    eng.load_adam_state(params_at(z, spec, prev + "adam_m/", with_Ws=False),
                        params_at(z, spec, prev + "adam_v/", with_Ws=False), steps_done=i)
    out = eng.forward_backward(z[pre + "X"], z[pre + "eps"], float(z[pre + "c"]), sigma_theta=spec["sigma_theta"], lam=spec["lam"])
    torch.cuda.synchronize()
    sc = out["scalars"].cpu().numpy()
    for j, k in enumerate(SCALARS):
      want = float(z[pre + k])
      assert abs(sc[j] - want) <= 1e-5 * max(abs(want), 1.0), (i, mode, k, sc[j], want)
    got_g = dict(flat_items(eng.export_grads()))
    gworst = 0.0
    for k, b in flat_items(params_at(z, spec, pre + "grad/")):
      gworst = max(gworst, rel_err(got_g[k], b))
      assert rel_err(got_g[k], b) < 2e-5, (i, mode, "grad", k, rel_err(got_g[k], b))
    eng.adam_step(spec["lr"])
    eng.sgd_prox_step(spec["Ws_lr"], float(z[pre + "t"]), do_prox=spec["lam"] > 0)
    torch.cuda.synchronize()
  finally:
    _lib.set_test_hooks(OIVAE_ENC=None, OIVAE_DEC=None)
  assert np.array_equal(eng.zero_mask.cpu().numpy(), z[pre + "zero_mask"]), (i, mode, "zero set")
  got_p = dict(flat_items(eng.export_params()))
  pworst, n_all, n_ok, g_all, g_ok = 0.0, 0, 0.0, 0, 0.0
  for k, b in flat_items(params_at(z, spec, pre + "post/")):
    pworst = max(pworst, rel_err(got_p[k], b))
    assert rel_err(got_p[k], b) < 1e-5, (i, mode, "param", k, rel_err(got_p[k], b))
    f, n = elementwise_within(got_p[k], b, rel=1e-5, floor=1e-3)
    n_all += n; n_ok += f * n
  for k, b in flat_items(params_at(z, spec, pre + "grad/")):
    f, n = elementwise_within(got_g[k], b, rel=1e-4, floor=1e-3)
    g_all += n; g_ok += f * n
  print("teacher-forced step %d (%s): worst gradient %.1e, worst weight %.1e of scale; elementwise: %.4f %% of %d "
        "weights within 1e-5 of themselves, %.3f %% of %d gradient elements within 1e-4 of themselves"
        % (i, mode, gworst, pworst, 100 * n_ok / n_all, n_all, 100 * g_ok / g_all, g_all))
  return dict(grad=gworst, weight=pworst, w_elem=n_ok / n_all, g_elem=g_ok / g_all)


@pytest.mark.parametrize("i", [12, 29])
def test_teacher_forced_steps_of_the_trained_run(i):
  """The per-step bar where the free-running comparison is loosest: steps 12 and 29 of c4_trained64 run from the
  REFERENCE's state after the step before -- its parameters and its Adam moments (make_golden.py `state_steps`) -- one
  step, no trajectory: scalars 1e-5, gradients 2e-5 and post-step weights 1e-5 of each tensor's scale, zero set exact,
  under the default kernels and under the fp32 matrix instructions.
  Elementwise, which the normwise bars do not show: of the post-step weights that are at least 1e-3 of their tensor's
  largest, at least 99 % are within 1e-5 of THEMSELVES, and the default kernels (two fp16 terms per operand) reach
  the fp32-instruction kernels' fraction to a tenth of a percentage point (measured: 99.48 % both at step 12, 99.75 %
  both at step 29 -- what is left is fp32 summation order, Adam's division by sqrt(v) included, not operand width;
  DESIGN.md section 4)."""
  print()
  d = _teacher_forced(i, "default")
  f = _teacher_forced(i, "fp32")
  assert d["w_elem"] >= 0.99 and d["w_elem"] >= f["w_elem"] - 1e-3, (d, f)
  assert d["g_elem"] >= f["g_elem"] - 2e-3, (d, f)
  assert d["grad"] <= 1.5 * max(f["grad"], 2e-6) and d["weight"] <= 1.5 * max(f["weight"], 2e-7), (d, f)
