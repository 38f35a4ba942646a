"""Pin the oracle (oracle/oivae_oracle.py) against the golden vectors recorded
from the real reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import oivae_oracle as O
import os
from tests._fixtures import GOLDEN, STEP_CONFIGS, flat_items, load, params_at, rel_err


# The torch build the golden vectors were recorded with: on it a live mismatch between the oracle and torch's CPU
# kernels is a regression of the ORACLE and fails; on any other torch (whose kernels may order these sums differently)
# the live sweeps skip and the golden vectors stay the pin.
RECORDED_WITH_TORCH = "2.10.0"


def _differs(what):
  msg = "oracle and torch %s disagree on %s" % (torch.__version__, what)
  if torch.__version__.startswith(RECORDED_WITH_TORCH):
    pytest.fail(msg)
  pytest.skip(msg)


def test_prox_vectors_bit_exact():
  z = np.load(os.path.join(GOLDEN, "prox_vectors.npz"))
  for H in z["Hs"]:
    W = z["H%d/W" % H]
    for key, t in (("out", float(z["H%d/t" % H])), ("out_small_t", 1e-4)):
      got, mask = O.proximal_step_reference_order(W, t)
      want = z["H%d/%s" % (H, key)]
      assert got.dtype == np.float32
      assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "H=%d %s" % (H, key)
      assert np.array_equal(mask, np.all(want == 0, axis=-1).astype(np.uint8))


def test_wide_prox_vectors_bit_exact():
  """Rows of 512 floats and more (tests/golden/prox_vectors_wide.npz, recorded from the reference): ATen's row sum
  cascades there -- a level flush every 16, 256 and 4096 steps of 32 floats -- and the oracle restates it."""
  z = np.load(os.path.join(GOLDEN, "prox_vectors_wide.npz"))
  for H in z["Hs"]:
    W = z["H%d/W" % H]
    keys = [("out", float(z["H%d/t" % H]))] + ([("out_small_t", 1e-4)] if "H%d/out_small_t" % H in z else [])
    for key, t in keys:
      got, mask = O.proximal_step_reference_order(W, t)
      want = z["H%d/%s" % (H, key)]
      assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "H=%d %s" % (H, key)
      assert np.array_equal(mask, np.all(want == 0, axis=-1).astype(np.uint8))


def test_row_sumsq_matches_torch_here():
  """Live check against torch's CPU kernel on this host (skipped silently on a
  host whose torch build orders the sum differently: the golden test above is
  the pin)."""
  rng = np.random.default_rng(0)
  for H in list(range(1, 70)) + [96, 100, 128, 200, 256, 300, 511, 512, 513, 543, 544, 1000, 4097, 8232, 131112]:
    a = rng.standard_normal((50, H)).astype(np.float32)
    want = torch.sum(torch.pow(torch.from_numpy(a).reshape(1, 50, H), 2), dim=2).numpy().reshape(-1)
    got = O.row_sumsq_reference_order(a)
    if not np.array_equal(got, want):
      _differs("the fp32 row sum at H=%d" % H)


@pytest.mark.parametrize("name", STEP_CONFIGS)
def test_oracle_tracks_reference(name):
  torch.set_num_threads(1)
  z, spec = load(name)
  prior_mu = prior_sigma = None
  tr = O.OracleTrainer(spec, params_at(z, spec, "init/"), lr=spec["lr"], Ws_lr=spec["Ws_lr"],
                       lam=spec["lam"], sigma_theta=spec["sigma_theta"],
                       prior_mu=prior_mu, prior_sigma=prior_sigma)
  for i in range(spec["n_steps"]):
    pre = "step%d/" % i
    info = tr.step(z[pre + "X"], z[pre + "eps"], float(z[pre + "c"]), float(z[pre + "t"]), want_pre_prox=True)
    for k in ("z_kl", "loglik_term", "logprob_theta", "logprob_W", "loss", "elbo"):
      want = float(z[pre + k])
      assert abs(info[k] - want) <= 2e-6 * max(abs(want), 1.0), (name, i, k, info[k], want)
    want_g = params_at(z, spec, pre + "grad/")
    got_g = dict(flat_items(info["grads"]))
    for k, b in flat_items(want_g):
      assert rel_err(got_g[k], b) < 1e-5, (name, i, "grad", k, rel_err(got_g[k], b))
    want_p = params_at(z, spec, pre + "post/")
    got_p = dict(flat_items(tr.model.export()))
    for k, b in flat_items(want_p):
      assert rel_err(got_p[k], b) < 1e-5, (name, i, "param", k, rel_err(got_p[k], b))
    # teacher-forced prox: same pre-prox bits -> same bits out, same zero set
    got, mask = O.proximal_step_reference_order(z[pre + "pre_prox_Ws"], float(z[pre + "t"]))
    assert np.array_equal(got.view(np.uint32), z[pre + "post/Ws"].view(np.uint32)), (name, i)
    assert np.array_equal(mask, z[pre + "zero_mask"])
    # free-running zero set (holds for these fixtures; not guaranteed in general)
    assert np.array_equal(info["zero_mask"], z[pre + "zero_mask"]), (name, i)


def test_oracle_tracks_reference_through_a_sparse_trained_run():
  """c4_trained64: 30 steps at the all-64 shape in which rows and whole groups die, revive and the generator and
  log_sigma move far from their initial values (tests/golden/make_golden.py).  Gradients, post-step parameters and
  pre-prox Ws are recorded at steps 0, 11, 12, 28 and 29 only; scalars and the zero set at every step."""
  torch.set_num_threads(1)
  z, spec = load("c4_trained64")
  tr = O.OracleTrainer(spec, params_at(z, spec, "init/"), lr=spec["lr"], Ws_lr=spec["Ws_lr"],
                       lam=spec["lam"], sigma_theta=spec["sigma_theta"])
  dead_seen = 0
  for i in range(spec["n_steps"]):
    pre = "step%d/" % i
    full = (pre + "post/Ws") in z.files
    info = tr.step(z[pre + "X"], z[pre + "eps"], float(z[pre + "c"]), float(z[pre + "t"]), want_grads=full)
    for k in ("z_kl", "loglik_term", "logprob_theta", "logprob_W", "loss", "elbo"):
      want = float(z[pre + k])
      assert abs(info[k] - want) <= 5e-6 * max(abs(want), 1.0), (i, k, info[k], want)
    assert np.array_equal(info["zero_mask"], z[pre + "zero_mask"]), i
    dead_seen = max(dead_seen, int(z[pre + "zero_mask"].sum()))
    if full:
      got_g = dict(flat_items(info["grads"]))
      for k, b in flat_items(params_at(z, spec, pre + "grad/")):
        assert rel_err(got_g[k], b) < 2e-5, (i, "grad", k, rel_err(got_g[k], b))
      got_p = dict(flat_items(tr.model.export()))
      for k, b in flat_items(params_at(z, spec, pre + "post/")):
        assert rel_err(got_p[k], b) < 1e-5, (i, "param", k, rel_err(got_p[k], b))
  assert dead_seen >= 140 and z["step1/zero_mask"][3].all()      # the fixture is what it claims to be


@pytest.mark.parametrize("i", [12, 29])
def test_oracle_teacher_forced_from_the_reference_state(i):
  """Steps 12 and 29 of c4_trained64 run from the reference's OWN state after the step before (parameters and Adam's
  moments, recorded by make_golden.py `state_steps`): one step, no trajectory -- the per-step bar of the north star."""
  torch.set_num_threads(1)
  z, spec = load("c4_trained64")
  prev, pre = "step%d/" % (i - 1), "step%d/" % i
  tr = O.OracleTrainer(spec, params_at(z, spec, prev + "post/"), lr=spec["lr"], Ws_lr=spec["Ws_lr"],
                       lam=spec["lam"], sigma_theta=spec["sigma_theta"])
  tr.load_adam_state(params_at(z, spec, prev + "adam_m/", with_Ws=False),
                     params_at(z, spec, prev + "adam_v/", with_Ws=False), steps_done=i)
  info = tr.step(z[pre + "X"], z[pre + "eps"], float(z[pre + "c"]), float(z[pre + "t"]))
  for k in ("z_kl", "loglik_term", "logprob_theta", "logprob_W", "loss", "elbo"):
    want = float(z[pre + k])
    assert abs(info[k] - want) <= 2e-6 * max(abs(want), 1.0), (i, k, info[k], want)
  got_g = dict(flat_items(info["grads"]))
  for k, b in flat_items(params_at(z, spec, pre + "grad/")):
    assert rel_err(got_g[k], b) < 1e-5, (i, "grad", k, rel_err(got_g[k], b))
  got_p = dict(flat_items(tr.model.export()))
  for k, b in flat_items(params_at(z, spec, pre + "post/")):
    assert rel_err(got_p[k], b) < 2e-6, (i, "param", k, rel_err(got_p[k], b))
  assert np.array_equal(info["zero_mask"], z[pre + "zero_mask"]), i


@pytest.mark.parametrize("fixture,least", [("fls_prox_vectors.npz", 200), ("fls_prox_vectors_tall.npz", 27)])
def test_column_prox_oracle_matches_reference_bits(fixture, least):
  """FirstLayerSparseDecoder.proximal_step (lib/models.py:65-71): the oracle's column-sum order, MKL root and
  two-rounding scale against 228 cases recorded from the reference (tests/golden/make_golden_fls.py) and 27 more with
  maps of 272 to 9000 rows, where the column sum's cascade uses its third and fourth level (`... tall`)."""
  from oracle import prox_oracle
  d = np.load(os.path.join(GOLDEN, fixture))
  assert len(d["names"]) >= least
  for name in d["names"]:
    ci = name.split("_")[0]
    W, dims, t = d[ci + "/W"], d[name + "/dims"], float(d[name + "/t"])
    got, mask, norms = prox_oracle.proximal_step_cols(W, dims, t)
    want = d[name + "/out"]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
    off = np.concatenate([[0], np.cumsum(dims)])
    wmask = np.stack([(want[off[g]:off[g + 1]] == 0).all(0) for g in range(len(dims))]).astype(np.uint8)
    assert np.array_equal(mask, wmask), name
    # group_lasso_penalty (lib/models.py:59-63) is the sum of the column norms the step used
    assert abs(float(norms.astype(np.float64).sum()) - float(d[name + "/pen"])) <= 2e-6 * max(1.0, abs(float(d[name + "/pen"])))


def test_prox_oracles_match_torch_here_on_a_sweep_of_shapes():
  """Live sweep against torch's CPU kernels on this host (the golden vectors are the pin; this is the net that
  caught the column sum's missing cascade levels): the reference's two proximal steps written with the same torch
  calls (lib/models.py:65-71, 106-111) against the C oracle, widths on every side of the 8 / 32 / 512 / 8192 boundaries
  of the row sum and rows x columns on every side of the 4 / 8 / 32-column and 16 / 256 / 4096-row boundaries of the
  column sum.  Skipped on a host whose torch orders these sums differently."""
  from oracle import prox_oracle
  torch.set_num_threads(1)
  rng = np.random.default_rng(2)
  for H in list(range(1, 70)) + list(range(70, 1100, 37)) + [511, 512, 513, 543, 544, 4095, 4096, 4104, 8191, 8192, 8224, 131104]:
    W = (rng.standard_normal((2, 3, H)) * rng.uniform(0.1, 2.0)).astype(np.float32)
    t = float(0.4 * np.sqrt(H))
    Wt = torch.from_numpy(W.copy())
    nn = torch.sqrt(torch.sum(torch.pow(Wt, 2), dim=2, keepdim=True))
    Wt.div_(torch.clamp(nn, min=1e-16)); Wt.mul_(torch.clamp(nn - t, min=0))
    got, _, _ = prox_oracle.proximal_step(W, t)
    if not np.array_equal(got.view(np.uint32), Wt.numpy().view(np.uint32)):
      _differs("the row form of the proximal step at H=%d" % H)
  for Hg in [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 64, 100, 255, 256, 257, 271, 272, 273, 300, 511, 512, 1000, 4095, 4096, 4112, 4113]:
    for K in [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 40, 64, 65, 100]:
      if Hg * K >= 32768 or (K == 1 and Hg >= 512):
        continue
      W = (rng.standard_normal((Hg, K)) * rng.uniform(0.1, 2.0)).astype(np.float32)
      t = float(0.4 * np.sqrt(Hg))
      Wt = torch.from_numpy(W.copy())
      nn = torch.sqrt(torch.sum(torch.pow(Wt, 2), dim=0, keepdim=True))
      Wt.div_(torch.clamp(nn, min=1e-16)); Wt.mul_(torch.clamp(nn - t, min=0))
      got, _, _ = prox_oracle.proximal_step_cols(W, [Hg], t)
      if not np.array_equal(got.view(np.uint32), Wt.numpy().view(np.uint32)):
        _differs("the column form of the proximal step at H_g=%d, K=%d" % (Hg, K))
