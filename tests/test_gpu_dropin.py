"""GPU tests of the drop-in layer: models built exactly as the reference's
scripts build them (bars_data_oivae.py:66-108,212-253), driven through
`OIVAE.step` with caller-owned torch.optim optimizers and with the engine-native
ones, compared with the golden vectors recorded from the real reference."""
import numpy as np
import pytest
import torch

from tests._fixtures import flat_items, load, params_at, rel_err

pytestmark = pytest.mark.gpu

SCALARS = ("z_kl", "loglik_term", "logprob_theta", "logprob_W", "loss", "elbo")
MOCAP_DIMS = [3, 3, 3, 3, 3, 3, 3, 2, 3, 1, 1, 2, 1, 2, 2, 3, 1, 1, 2, 1, 2, 3, 1, 2, 1, 3, 1, 2, 1]


def _build(spec, holder):
  import oi_vae_amd as oi
  D, K, H, M, typ, dims = spec["D"], spec["K"], spec["H"], spec.get("M", 0), spec["gen_type"], spec["group_dims"]
  inference_net = oi.NormalNet(
    mu_net=torch.nn.Sequential(torch.nn.Linear(D, K)),
    sigma_net=torch.nn.Sequential(torch.nn.Linear(D, K), oi.Lambda(torch.exp),
                                  oi.Lambda(lambda x: x * holder["c"] + 1e-3)))

  def make_gen(out_dim):
    log_sigma = torch.log(1e-2 * torch.ones(out_dim)).requires_grad_(True)
    if typ == "linear":
      mu_net = torch.nn.Linear(H, out_dim)
    elif typ == "tanh_linear":
      mu_net = torch.nn.Sequential(torch.nn.Tanh(), torch.nn.Linear(H, out_dim))
    else:
      mu_net = torch.nn.Sequential(torch.nn.Linear(H, M), torch.nn.Tanh(), torch.nn.Linear(M, out_dim))
    return oi.NormalNet(mu_net=mu_net,
                        sigma_net=oi.Lambda(lambda x, ls: torch.exp(ls.expand(x.size(0), -1)) + 1e-3,
                                            extra_args=(log_sigma,)))
  gen = oi.BayesianGroupLassoGenerator([make_gen(d) for d in dims], group_input_dim=H, dim_z=K)
  return inference_net, gen


def _load_into(inference_net, gen, p, typ):
  with torch.no_grad():
    inference_net.mu_net[0].weight.copy_(torch.from_numpy(p["enc_Wmu"])); inference_net.mu_net[0].bias.copy_(torch.from_numpy(p["enc_bmu"]))
    inference_net.sigma_net[0].weight.copy_(torch.from_numpy(p["enc_Wsig"])); inference_net.sigma_net[0].bias.copy_(torch.from_numpy(p["enc_bsig"]))
    for g, gg in enumerate(gen.group_generators):
      mods = list(gg.mu_net) if isinstance(gg.mu_net, torch.nn.Sequential) else [gg.mu_net]
      lin2 = mods[-1]
      lin2.weight.copy_(torch.from_numpy(p["gen_W2"][g])); lin2.bias.copy_(torch.from_numpy(p["gen_b2"][g]))
      if typ == "mlp":
        mods[0].weight.copy_(torch.from_numpy(p["gen_W1"][g])); mods[0].bias.copy_(torch.from_numpy(p["gen_b1"][g]))
      gg.sigma_net.extra_args[0].copy_(torch.from_numpy(p["log_sigma"][g]))
    gen.Ws.copy_(torch.from_numpy(p["Ws"]))


@pytest.mark.parametrize("name,native", [("c1_bars_k8", False), ("c1_bars_k8", True), ("c1_bars_k16", True),
                                         ("c2_mocap", False), ("c3_meg_like", False), ("c4_tiny32", True),
                                         ("c4_tiny64", True), ("c4_tiny64", False), ("c4_mixed", False),
                                         ("c5_wide", True), ("c1_bars_k16_strong", True)])
def test_scripts_style_training_matches_reference(name, native):
  import oi_vae_amd as oi
  z, spec = load(name)
  holder = {"c": 0.1}
  inference_net, gen = _build(spec, holder)
  _load_into(inference_net, gen, params_at(z, spec, "init/"), spec["gen_type"])
  adam_groups = [
    {"params": inference_net.parameters(), "lr": spec["lr"]},
    {"params": gen.group_generators_parameters(), "lr": spec["lr"]},
    {"params": [g.sigma_net.extra_args[0] for g in gen.group_generators], "lr": spec["lr"]}]
  sgd_groups = [{"params": [gen.Ws], "lr": spec["Ws_lr"], "momentum": 0}]
  if native:
    optimizer, optimizer_Ws = oi.optim.Adam(adam_groups), oi.optim.SGD(sgd_groups)
  else:
    optimizer, optimizer_Ws = torch.optim.Adam(adam_groups), torch.optim.SGD(sgd_groups)
  rows = z["step0/X"].shape[0] if spec["prior_rows"] == "B" else 1
  prior_z = oi.Normal(torch.zeros(rows, spec["K"]), torch.ones(rows, spec["K"]))
  vae = oi.OIVAE(inference_model=inference_net, generative_model=gen, prior_z=prior_z,
                 prior_theta=oi.NormalPriorTheta(spec["sigma_theta"]), lam=spec["lam"],
                 optimizers=[optimizer, optimizer_Ws])
  assert gen.Ws.is_cuda and inference_net.mu_net[0].weight.is_cuda
  for i in range(spec["n_steps"]):
    pre = "step%d/" % i
    holder["c"] = float(z[pre + "c"])
    X = torch.from_numpy(z[pre + "X"])
    S = z[pre + "eps"].shape[0]
    info = vae.step(X=X, prox_step_size=float(z[pre + "t"]), mc_samples=S, eps=torch.from_numpy(z[pre + "eps"]))
    assert sorted(info) == sorted(["q_z", "z_kl", "z_sample", "loglik_term", "logprob_theta", "logprob_W", "loss", "elbo"])
    for k in SCALARS:
      want = float(z[pre + k])
      assert info[k].dim() == 0
      assert abs(info[k].item() - want) <= 1e-5 * max(abs(want), 1.0), (name, i, k, info[k].item(), want)
    assert info["z_sample"].shape == (S * X.shape[0], spec["K"])
    assert rel_err(info["q_z"].mu.cpu().numpy(), z[pre + "mu_q"]) < 1e-5
    # gradients are left populated on the caller's tensors (lib/oivae.py:73-75)
    assert rel_err(gen.Ws.grad.cpu().numpy(), z[pre + "grad/Ws"]) < 2e-5
    assert rel_err(inference_net.mu_net[0].weight.grad.cpu().numpy(), z[pre + "grad/enc_Wmu"]) < 2e-5
    want_p = params_at(z, spec, pre + "post/")
    assert rel_err(gen.Ws.data.cpu().numpy(), want_p["Ws"]) < 1e-5, (name, i)
    assert rel_err(inference_net.sigma_net[0].weight.data.cpu().numpy(), want_p["enc_Wsig"]) < 1e-5
    ga, gb = min(3, len(gen.group_generators) - 1), min(2, len(gen.group_generators) - 1)
    lin2 = gen.group_generators[ga].mu_net
    lin2 = list(lin2)[-1] if isinstance(lin2, torch.nn.Sequential) else lin2
    assert rel_err(lin2.weight.data.cpu().numpy(), want_p["gen_W2"][ga]) < 1e-5
    assert rel_err(gen.group_generators[gb].sigma_net.extra_args[0].data.cpu().numpy(), want_p["log_sigma"][gb]) < 1e-5
    zero = (gen.Ws.data.abs().sum(dim=2) == 0).cpu().numpy().astype(np.uint8)
    assert np.array_equal(zero, z[pre + "zero_mask"]), (name, i, "zero set")


def test_default_eps_follows_the_cpu_generator_like_the_reference():
  import oi_vae_amd as oi
  z, spec = load("c1_bars_k8")
  outs = []
  for explicit in (False, True):
    holder = {"c": 0.1}
    inference_net, gen = _build(spec, holder)
    _load_into(inference_net, gen, params_at(z, spec, "init/"), spec["gen_type"])
    vae = oi.OIVAE(inference_net, gen, oi.Normal(torch.zeros(1, 8), torch.ones(1, 8)), oi.NormalPriorTheta(1.0), 1.0,
                   [torch.optim.SGD([gen.Ws], lr=1e-4)])
    X = torch.from_numpy(z["step0/X"])
    torch.manual_seed(123)
    if explicit:
      eps = torch.stack([torch.randn(64, 8) for _ in range(3)])
      info = vae.step(X, 1e-4, 3, eps=eps)
    else:
      info = vae.step(X, 1e-4, 3)
    outs.append(info["z_sample"].cpu().numpy())
  assert np.array_equal(outs[0], outs[1])


def test_inference_time_calls_work_on_adopted_model():
  import oi_vae_amd as oi
  z, spec = load("c2_mocap")
  holder = {"c": 1.0}
  inference_net, gen = _build(spec, holder)
  oi.OIVAE(inference_net, gen, oi.Normal(torch.zeros(1, 8), torch.ones(1, 8)), oi.NormalPriorTheta(1.0), 1.0, [])
  X = torch.from_numpy(z["step0/X"]).cuda()
  recon = gen(inference_net(X).mu)
  assert recon.mu.shape == (64, 59) and recon.sigma.shape == (64, 59)
  assert recon.sample().shape == (64, 59)
  assert gen.connectivity().shape == (29, 8)
  gen.proximal_step(100.0)
  assert float(gen.Ws.data.abs().sum()) == 0.0


@pytest.mark.parametrize("name", ["c2_mocap", "c4_tiny32"])
def test_evaluate_gives_the_steps_terms_without_side_effects(name):
  """Held-out evaluation (stubbed in the reference, meg_oi_vae.py:258-260): the terms of the
  golden step, with parameters, gradients and optimizer state left alone."""
  import oi_vae_amd as oi
  z, spec = load(name)
  holder = {"c": float(z["step0/c"])}
  inference_net, gen = _build(spec, holder)
  _load_into(inference_net, gen, params_at(z, spec, "init/"), spec["gen_type"])
  opt = torch.optim.SGD([gen.Ws], lr=spec["Ws_lr"])
  rows = z["step0/X"].shape[0] if spec["prior_rows"] == "B" else 1
  vae = oi.OIVAE(inference_net, gen, oi.Normal(torch.zeros(rows, spec["K"]), torch.ones(rows, spec["K"])),
                 oi.NormalPriorTheta(spec["sigma_theta"]), spec["lam"], [opt])
  before = vae.engine.flat_params.clone()
  grads_before = vae.engine.flat_grads.clone()
  X, eps = torch.from_numpy(z["step0/X"]), torch.from_numpy(z["step0/eps"])
  for _ in range(2):
    info = vae.evaluate(X, eps.shape[0], eps=eps)
    for k in SCALARS:
      want = float(z["step0/" + k])
      assert abs(info[k].item() - want) <= 1e-5 * max(abs(want), 1.0), (k, info[k].item(), want)
    assert rel_err(info["q_z"].mu.cpu().numpy(), z["step0/mu_q"]) < 1e-5
  assert torch.equal(vae.engine.flat_params, before)
  assert torch.equal(vae.engine.flat_grads, grads_before)
  # .grad is the view of the engine's gradient buffer that adopt() bound once: still there, still untouched
  assert gen.Ws.grad.data_ptr() == vae.engine.g["Ws"].data_ptr() and float(gen.Ws.grad.abs().sum()) == 0.0
  # and a training step afterwards is still the golden one
  info = vae.step(X, float(z["step0/t"]), eps.shape[0], eps=eps)
  assert rel_err(gen.Ws.grad.cpu().numpy(), z["step0/grad/Ws"]) < 2e-5


def test_autoencoder_variant_driver_trains_and_sparsifies(tmp_path):
  """scripts/data_vs_test_loss_autoencoder.py (counterpart of the reference's driver of the same name) for a few hundred
  epochs at two training-set sizes: the reconstruction loss falls, and with lambda > 0 the proximal kernel zeroes
  connections exactly while lambda = 0 zeroes none."""
  import importlib.util
  import json
  import os
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  spec = importlib.util.spec_from_file_location("ae_driver", os.path.join(root, "scripts", "data_vs_test_loss_autoencoder.py"))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  out = tmp_path / "ae.json"
  res = mod.main(["--epochs", "300", "--train-sizes", "64", "--lams", "0.0", "0.1", "--test-samples", "256", "--out", str(out)])
  saved = json.load(open(out))
  assert len(saved["results"]) == 2 and saved["lams"] == [0.0, 0.1]
  by_lam = {r["lam"]: r for r in res}
  for r in res:
    assert r["train_loss"][-1] < 0.5 * r["train_loss"][0], (r["lam"], r["train_loss"][0], r["train_loss"][-1])
    assert all(v == v for v in r["test_loss"])
  assert by_lam[0.0]["zero_connections"] == 0
  assert by_lam[0.1]["zero_connections"] > 0
