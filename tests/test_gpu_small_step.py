"""The one-launch step for the scripts' own small shapes (csrc/small_step.hip, oivae_small_step_f32) against the golden
vectors recorded from the real reference: the same bars / mocap / wide fixtures and the same bars the multi-launch path
is held to (tests/test_gpu_parity.py) -- scalars 1e-5 relative, mu_q / sig_q 1e-5, every gradient 2e-5 and every
post-step weight 1e-5 of its tensor's scale, the zero set exact at every step, free-running over all recorded steps
(c flips, a ragged last batch, S = 10) -- plus: same bits run to run, and agreement with the multi-launch path."""
import numpy as np
import pytest
import torch

from tests._fixtures import flat_items, load, params_at, rel_err

pytestmark = pytest.mark.gpu
SCALARS = ("z_kl", "loglik_term", "logprob_theta", "logprob_W", "loss", "elbo")
SMALL = ["c1_bars_k8", "c1_bars_k16", "c1_bars_k16_strong", "c2_mocap"]   # (c5_wide: 35 k parameters do not fit the LDS stage)


def _engine(spec):
  from oi_vae_amd import Engine, ModelSpec
  return Engine(ModelSpec(D=spec["D"], K=spec["K"], H=spec["H"], group_dims=spec["group_dims"],
                          gen_type=spec["gen_type"], M=spec.get("M", 0)), device="cuda:0")


@pytest.mark.parametrize("name", SMALL)
def test_small_step_matches_reference_golden(name):
  z, spec = load(name)
  eng = _engine(spec)
  eng.load_params(params_at(z, spec, "init/"))
  for i in range(spec["n_steps"]):
    pre = "step%d/" % i
    X, eps = z[pre + "X"], z[pre + "eps"]
    assert eng.small_step_applies(X.shape[0], eps.shape[0]), name
    prior = None
    if spec["prior_rows"] == "B":
      prior = (torch.zeros(X.shape[0], spec["K"]), torch.ones(X.shape[0], spec["K"]))
    out = eng.small_step(X, eps, float(z[pre + "c"]), float(z[pre + "t"]), lr=spec["lr"], Ws_lr=spec["Ws_lr"],
                         do_prox=spec["lam"] > 0, sigma_theta=spec["sigma_theta"], lam=spec["lam"],
                         prior_mu=prior[0] if prior else None, prior_sigma=prior[1] if prior else None)
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
    got_p = dict(flat_items(eng.export_params()))
    for k, b in flat_items(params_at(z, spec, pre + "post/")):
      assert rel_err(got_p[k], b) < 1e-5, (name, i, "param", k, rel_err(got_p[k], b))
    assert np.array_equal(eng.zero_mask.cpu().numpy(), z[pre + "zero_mask"]), (name, i, "zero set")


@pytest.mark.parametrize("name", ["c2_mocap", "c1_bars_k16_strong"])
def test_small_step_is_bit_reproducible_and_agrees_with_the_multi_launch_path(name):
  z, spec = load(name)
  X, eps, c, t = z["step0/X"], z["step0/eps"], float(z["step0/c"]), float(z["step0/t"])
  runs = []
  for rep in range(2):
    eng = _engine(spec)
    eng.load_params(params_at(z, spec, "init/"))
    out = eng.small_step(X, eps, c, t, lr=spec["lr"], Ws_lr=spec["Ws_lr"], sigma_theta=spec["sigma_theta"], lam=spec["lam"])
    torch.cuda.synchronize()
    runs.append((eng.flat_grads.clone(), eng.flat_params.clone(), out["scalars"].clone(), out["z"].clone()))
  for a, b in zip(runs[0], runs[1]):
    assert torch.equal(a, b)              # no float atomics anywhere in this path
  ref = _engine(spec)
  ref.load_params(params_at(z, spec, "init/"))
  o = ref.forward_backward(X, eps, c, sigma_theta=spec["sigma_theta"], lam=spec["lam"])
  ref.adam_step(spec["lr"])
  ref.sgd_prox_step(spec["Ws_lr"], t, do_prox=spec["lam"] > 0)
  torch.cuda.synchronize()
  assert ((runs[0][2][:6] - o["scalars"][:6]).abs() / o["scalars"][:6].abs().clamp(min=1)).max().item() < 1e-5
  for nme in ref.seg_off:
    lo, n = ref.seg_off[nme], ref.seg_numel[nme]
    if n:
      g0, g1 = runs[0][0][lo:lo + n], ref.flat_grads[lo:lo + n]
      assert float((g0 - g1).abs().max()) <= 2e-5 * float(g1.abs().max()) + 1e-30, nme
      p0, p1 = runs[0][1][lo:lo + n], ref.flat_params[lo:lo + n]
      assert float((p0 - p1).abs().max()) <= 1e-5 * float(p1.abs().max()) + 1e-30, nme
  assert torch.equal(runs[0][1][ref.seg_off["Ws"]:] == 0, ref.flat_params[ref.seg_off["Ws"]:] == 0)


def test_small_step_refuses_what_it_does_not_cover():
  from oi_vae_amd import Engine, ModelSpec, _lib
  eng = Engine(ModelSpec(D=128, K=64, H=64, M=64, group_dims=[64, 64], gen_type="mlp"), device="cuda:0")
  assert not eng.small_step_applies(64, 1)
  with pytest.raises(_lib.OivaeError):
    eng.small_step(torch.rand(64, 128), torch.randn(1, 64, 64), 1.0, 1e-4, lr=1e-3, Ws_lr=1e-4)


def test_small_step_ticket_survives_a_foreign_workspace_and_interleaved_calls():
  """The kernel's "last workgroup" ticket lives in the workspace.  The entry point zeroes it first unless the caller
  vouches for it (ticket_is_zero, include/oivae_hip.h); the engine vouches only when the same workspace went through a
  completed small step of the same B and S with nothing in between.  Here: a workspace full of 0xff, then steps that
  alternate with the multi-launch path at another batch size (which re-carves the same allocation) -- every small step
  must give the bits of a fresh engine doing the same steps without the shortcut."""
  z, spec = load("c2_mocap")
  X, eps, c, t = z["step0/X"], z["step0/eps"], float(z["step0/c"]), float(z["step0/t"])
  kw = dict(lr=spec["lr"], Ws_lr=spec["Ws_lr"], sigma_theta=spec["sigma_theta"], lam=spec["lam"])

  def run(shortcut):
    eng = _engine(spec)
    eng.load_params(params_at(z, spec, "init/"))
    eng._workspace(X.shape[0], eps.shape[0]).fill_(0xff)          # a foreign workspace: the ticket word is garbage
    outs = []
    for rep in range(6):
      if not shortcut:
        eng._small_ticket = None                                    # always the memset node
      if rep == 3:   # another path, another batch size, the same allocation
        eng.forward_backward(X[:37], eps[:, :37], c)
      out = eng.small_step(X, eps, c, t, **kw)
      outs.append((out["scalars"].clone(), eng.flat_params.clone()))
    torch.cuda.synchronize()
    return outs, eng

  a, eng = run(True)
  b, _ = run(False)
  assert eng._small_ticket is not None                              # the shortcut was armed at the end
  for (sa, pa), (sb, pb) in zip(a, b):
    assert torch.equal(sa, sb) and torch.equal(pa, pb)
