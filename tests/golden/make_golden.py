#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Run in the build container only (the reference lives at /root/reference, which
does not exist on the GPU box):

    python tests/golden/make_golden.py

It imports the reference's ``lib`` package, builds small models exactly as the
scripts wire them (bars_data_oivae.py:66-108,212-253; mo_mo_mocap_oivae.py:
172-291; meg_oi_vae.py:147-222), drives the unmodified ``OIVAE.step``
(lib/oivae.py:42-90) for a few steps and records inputs and outputs as ``.npz``
files.  Nothing of the reference's source is written anywhere; the fixtures hold
numbers only.

Harness-side shims (not changes to the reference):
  * ``PriorZ``: lib/oivae.py:50 hands a Normal to Tensor.expand_as, which
    torch >= 0.4 rejects; prior_z is caller-supplied, so the harness supplies a
    Normal subclass whose expand_as takes the Normal's size.
  * ``RecordingGenerator``: subclass of BayesianGroupLassoGenerator that copies
    Ws just before proximal_step runs, so the prox kernel can be checked
    teacher-forced (same pre-prox bits).
  * eps: Normal.sample draws torch.randn from the CPU default generator
    (lib/distributions.py:36-38) and nothing else in step() consumes RNG, so the
    harness seeds the generator before each step and re-draws the same numbers.
"""
import json
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from lib.distributions import Normal  # noqa: E402
from lib.models import BayesianGroupLassoGenerator, NormalNet  # noqa: E402
from lib.oivae import OIVAE, NormalPriorTheta  # noqa: E402
from lib.utils import Lambda  # noqa: E402
from lib.bars_data import sample_one_bar_image  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


class PriorZ(Normal):
  def expand_as(self, other):
    return Normal(self.mu.expand(other.size()), self.sigma.expand(other.size()))


class RecordingGenerator(BayesianGroupLassoGenerator):
  def proximal_step(self, t):
    self.pre_prox = self.Ws.data.clone()
    super().proximal_step(t)


def build(spec, seed):
  torch.manual_seed(seed)
  D, K, H = spec["D"], spec["K"], spec["H"]
  dims = spec["group_dims"]
  typ = spec["gen_type"]
  M = spec.get("M", 0)
  holder = {"c": 0.1}
  inference_net = NormalNet(
    mu_net=torch.nn.Sequential(torch.nn.Linear(D, K)),
    sigma_net=torch.nn.Sequential(
      torch.nn.Linear(D, K),
      Lambda(torch.exp),
      Lambda(lambda x: x * holder["c"] + 1e-3)))

  def make_gen(out_dim):
    log_sigma = torch.log(1e-2 * torch.ones(out_dim)).requires_grad_(True)
    if typ == "linear":
      mu_net = torch.nn.Linear(H, out_dim)
    elif typ == "tanh_linear":
      mu_net = torch.nn.Sequential(torch.nn.Tanh(), torch.nn.Linear(H, out_dim))
    else:
      mu_net = torch.nn.Sequential(torch.nn.Linear(H, M), torch.nn.Tanh(), torch.nn.Linear(M, out_dim))
    return NormalNet(
      mu_net=mu_net,
      sigma_net=Lambda(lambda x, ls: torch.exp(ls.expand(x.size(0), -1)) + 1e-3, extra_args=(log_sigma,)))

  gen = RecordingGenerator([make_gen(d) for d in dims], group_input_dim=H, dim_z=K)
  return inference_net, gen, holder


def export_params(inference_net, gen, typ, grads=False):
  def val(p):
    if grads:
      return (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().copy()
    return p.detach().numpy().copy()
  out = {}
  out["enc_Wmu"] = val(inference_net.mu_net[0].weight)
  out["enc_bmu"] = val(inference_net.mu_net[0].bias)
  out["enc_Wsig"] = val(inference_net.sigma_net[0].weight)
  out["enc_bsig"] = val(inference_net.sigma_net[0].bias)
  for g, gg in enumerate(gen.group_generators):
    if typ == "linear":
      lin2 = gg.mu_net
    elif typ == "tanh_linear":
      lin2 = gg.mu_net[1]
    else:
      lin2 = gg.mu_net[2]
      out["gen_W1/%d" % g] = val(gg.mu_net[0].weight)
      out["gen_b1/%d" % g] = val(gg.mu_net[0].bias)
    out["gen_W2/%d" % g] = val(lin2.weight)
    out["gen_b2/%d" % g] = val(lin2.bias)
    out["log_sigma/%d" % g] = val(gg.sigma_net.extra_args[0])
  out["Ws"] = val(gen.Ws)
  return out


ONLY = set(sys.argv[1:])   # config names to (re)generate; none given = all


def adam_state(adam, inference_net, gen, typ, key):
  """torch.optim.Adam's per-parameter state (`exp_avg` / `exp_avg_sq`) in the layout of export_params."""
  def val(p):
    return adam.state[p][key].detach().numpy().copy()
  out = {}
  out["enc_Wmu"] = val(inference_net.mu_net[0].weight)
  out["enc_bmu"] = val(inference_net.mu_net[0].bias)
  out["enc_Wsig"] = val(inference_net.sigma_net[0].weight)
  out["enc_bsig"] = val(inference_net.sigma_net[0].bias)
  for g, gg in enumerate(gen.group_generators):
    lin2 = gg.mu_net if typ == "linear" else gg.mu_net[1] if typ == "tanh_linear" else gg.mu_net[2]
    if typ == "mlp":
      out["gen_W1/%d" % g] = val(gg.mu_net[0].weight)
      out["gen_b1/%d" % g] = val(gg.mu_net[0].bias)
    out["gen_W2/%d" % g] = val(lin2.weight)
    out["gen_b2/%d" % g] = val(lin2.bias)
    out["log_sigma/%d" % g] = val(gg.sigma_net.extra_args[0])
  return out


def run_config(name, spec, seed, steps, lr, Ws_lr, lam, prior_rows, sigma_theta=1.0, data="uniform",
               full_steps=None, prepare=None, state_steps=()):
  """steps: list of dicts(B=, S=, c=, t=).  full_steps: step indices whose gradients, post-step parameters and
  pre-prox Ws are recorded (None: every step); scalars, inputs and the zero mask are recorded for every step.
  state_steps: step indices after which Adam's moments are recorded too (stepN/adam_m/..., stepN/adam_v/...), so
  that step N+1 can be run TEACHER-FORCED from the reference's own complete state.
  prepare(inference_net, gen): harness-side preparation of the initial state (the state is the caller's)."""
  if ONLY and name not in ONLY:
    return
  inference_net, gen, holder = build(spec, seed)
  if prepare is not None:
    prepare(inference_net, gen)
  D, K = spec["D"], spec["K"]
  typ = spec["gen_type"]
  adam = torch.optim.Adam([
    {"params": inference_net.parameters(), "lr": lr},
    {"params": gen.group_generators_parameters(), "lr": lr},
    {"params": [g.sigma_net.extra_args[0] for g in gen.group_generators], "lr": lr}])
  sgd = torch.optim.SGD([{"params": [gen.Ws], "lr": Ws_lr, "momentum": 0}])
  B0 = steps[0]["B"]
  rows = B0 if prior_rows == "B" else 1
  prior = PriorZ(torch.zeros(rows, K), torch.ones(rows, K))
  vae = OIVAE(inference_model=inference_net, generative_model=gen, prior_z=prior,
              prior_theta=NormalPriorTheta(sigma_theta), lam=lam, optimizers=[adam, sgd])
  out = {"spec": json.dumps(dict(spec, lr=lr, Ws_lr=Ws_lr, lam=lam, sigma_theta=sigma_theta,
                                 prior_rows=prior_rows, n_steps=len(steps), seed=seed))}
  for k, v in export_params(inference_net, gen, typ).items():
    out["init/" + k] = v
  dg = torch.Generator().manual_seed(seed + 1000)
  for i, st in enumerate(steps):
    B, S, c, t = st["B"], st["S"], st["c"], st["t"]
    if data == "bars":
      X = torch.stack([sample_one_bar_image(8).view(-1) for _ in range(B)])
      X = X + 0.05 * torch.randn(X.size(), generator=dg)
    elif data == "randn":
      X = torch.randn(B, D, generator=dg)
    else:
      X = torch.rand(B, D, generator=dg)
    holder["c"] = c
    step_seed = 7000 + 13 * i + seed
    torch.manual_seed(step_seed)
    eps = torch.stack([torch.randn(B, K) for _ in range(S)])
    torch.manual_seed(step_seed)
    info = vae.step(X=X, prox_step_size=t, mc_samples=S)
    assert torch.equal(info["z_sample"].detach(),
                       torch.cat([info["q_z"].mu + info["q_z"].sigma * eps[s] for s in range(S)]).detach())
    pre = "step%d/" % i
    out[pre + "X"] = X.numpy().copy()
    out[pre + "eps"] = eps.numpy().copy()
    out[pre + "c"] = np.float64(c)
    out[pre + "t"] = np.float64(t)
    for k in ("z_kl", "loglik_term", "logprob_theta", "logprob_W", "loss", "elbo"):
      out[pre + k] = np.float32(info[k].detach().item())
    out[pre + "mu_q"] = info["q_z"].mu.detach().numpy().copy()
    out[pre + "sig_q"] = info["q_z"].sigma.detach().numpy().copy()
    if full_steps is None or i in full_steps:
      for k, v in export_params(inference_net, gen, typ, grads=True).items():
        out[pre + "grad/" + k] = v
      for k, v in export_params(inference_net, gen, typ).items():
        out[pre + "post/" + k] = v
      out[pre + "pre_prox_Ws"] = gen.pre_prox.numpy().copy()
    if i in state_steps:
      for key, tag in (("exp_avg", "adam_m/"), ("exp_avg_sq", "adam_v/")):
        for k, v in adam_state(adam, inference_net, gen, typ, key).items():
          out[pre + tag + k] = v
    out[pre + "zero_mask"] = (gen.Ws.data.abs().sum(dim=2) == 0).numpy().astype(np.uint8)
    print(name, "step", i, "elbo", float(info["elbo"]), "zero rows", int(out[pre + "zero_mask"].sum()),
          "of", out[pre + "zero_mask"].size)
  path = os.path.join(HERE, name + ".npz")
  np.savez_compressed(path, **out)
  print("wrote", path, os.path.getsize(path), "bytes")


def reference_prox(Ws, t):
  """Run the reference's proximal_step (lib/models.py:106-111) on given bits."""
  G, K, H = Ws.shape
  gen = BayesianGroupLassoGenerator([None] * G, group_input_dim=H, dim_z=K)
  gen.Ws = torch.from_numpy(Ws.copy()).requires_grad_(True)
  gen.proximal_step(t)
  return gen.Ws.data.numpy().copy()


def prox_vectors():
  """Adversarial rows for the proximal step: norm exactly t, one ulp either
  side, zero rows, denormal rows, huge rows; many H so every branch of the
  reduction order is visited."""
  rng = np.random.default_rng(1234)
  out = {}
  Hs = [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 17, 24, 31, 32, 33, 40, 48, 63, 64, 65, 72, 96, 100, 128, 200, 256]
  out["Hs"] = np.array(Hs, np.int32)
  for H in Hs:
    G, K = 6, 40
    W = rng.standard_normal((G, K, H)).astype(np.float32)
    t = np.float32(0.75 * math.sqrt(H))
    # scale some rows so their norm lands within a few ulps of t
    for g in range(G):
      for k in range(0, K, 4):
        n = np.sqrt(np.sum(W[g, k].astype(np.float64) ** 2))
        W[g, k] = (W[g, k].astype(np.float64) * (float(t) / n)).astype(np.float32)
      for k in range(1, K, 8):
        n = np.sqrt(np.sum(W[g, k].astype(np.float64) ** 2))
        W[g, k] = (W[g, k].astype(np.float64) * (float(np.nextafter(t, np.float32(9))) / n)).astype(np.float32)
    W[0, 2] = 0
    W[1, 3] = -0.0
    W[2, 2] = np.float32(1e-42)           # denormal elements
    W[3, 2] = np.float32(1e-30)
    W[4, 2] = np.float32(3e18)            # square overflows towards inf at large H? (9e36 * H stays finite)
    W[5, 2, 0] = np.float32(1e-20)
    if H > 1:
      W[5, 2, 1:] = 0
    out["H%d/W" % H] = W
    out["H%d/t" % H] = np.float64(t)
    out["H%d/out" % H] = reference_prox(W, float(t))
    # second threshold: tiny t as in the scripts (Ws_lr*lam = 1e-4)
    out["H%d/out_small_t" % H] = reference_prox(W, 1e-4)
  path = os.path.join(HERE, "prox_vectors.npz")
  np.savez_compressed(path, **out)
  print("wrote", path, os.path.getsize(path), "bytes")


def prox_vectors_wide():
  """Rows of 512 floats and more, where ATen's row sum cascades (every 16 steps of 32 floats a level-0 flush, every
  256 a level-1 flush, every 4096 a level-2 flush): widths on both sides of every boundary, rows at the threshold,
  zero / tiny rows.  A file of its own (prox_vectors.npz stays as recorded)."""
  rng = np.random.default_rng(4321)
  out = {}
  Hs = [512, 513, 520, 544, 1000, 1031, 4100, 8200, 131112]
  out["Hs"] = np.array(Hs, np.int32)
  for H in Hs:
    big = H > 100000   # (one row, one threshold: the file stays small)
    G, K = (1, 1) if big else (1, 5)
    W = (rng.standard_normal((G, K, H)) * (0.2 + rng.random((G, K, 1)))).astype(np.float32)
    t = np.float32(0.7 * math.sqrt(H))
    if not big:
      n = np.sqrt(np.sum(W[0, 0].astype(np.float64) ** 2))
      W[0, 0] = (W[0, 0].astype(np.float64) * (float(t) / n)).astype(np.float32)
      n = np.sqrt(np.sum(W[0, 1].astype(np.float64) ** 2))
      W[0, 1] = (W[0, 1].astype(np.float64) * (float(np.nextafter(t, np.float32(1e9))) / n)).astype(np.float32)
      W[0, 2] = 0
      W[0, 3] = np.float32(1e-30)
    out["H%d/W" % H] = W
    out["H%d/t" % H] = np.float64(t)
    out["H%d/out" % H] = reference_prox(W, float(0.5 * t if big else t))
    if big:
      out["H%d/t" % H] = np.float64(0.5 * t)
    else:
      out["H%d/out_small_t" % H] = reference_prox(W, 1e-4)
  path = os.path.join(HERE, "prox_vectors_wide.npz")
  np.savez_compressed(path, **out)
  print("wrote", path, os.path.getsize(path), "bytes")


MOCAP_DIMS = [3, 3, 3, 3, 3, 3, 3, 2, 3, 1, 1, 2, 1, 2, 2, 3, 1, 1, 2, 1, 2, 3, 1, 2, 1, 3, 1, 2, 1]


def main():
  torch.set_num_threads(1)
  # C1 bars (bars_data_oivae.py), script K=8 and BASELINE K=16; c flips 0.1 -> 1.
  for K in (8, 16):
    run_config("c1_bars_k%d" % K,
               dict(D=64, K=K, H=1, M=0, gen_type="linear", group_dims=[8] * 8), seed=K,
               steps=[dict(B=64, S=1, c=0.1, t=1e-4)] * 3 + [dict(B=64, S=1, c=1.0, t=1e-4)] * 3,
               lr=1e-2, Ws_lr=1e-4, lam=1, prior_rows="B", data="bars")
  # C1 with a strong threshold so rows actually die.
  run_config("c1_bars_k16_strong",
             dict(D=64, K=16, H=1, M=0, gen_type="linear", group_dims=[8] * 8), seed=3,
             steps=[dict(B=64, S=1, c=0.1, t=0.4)] * 4,
             lr=1e-2, Ws_lr=1e-4, lam=1, prior_rows="B", data="bars")
  # C2 mocap shapes: 29 ragged joints, Tanh->Linear, S=10, ragged last batch.
  run_config("c2_mocap",
             dict(D=59, K=8, H=8, M=0, gen_type="tanh_linear", group_dims=MOCAP_DIMS), seed=5,
             steps=[dict(B=64, S=10, c=0.1, t=1e-4)] * 2 + [dict(B=64, S=10, c=1.0, t=1e-4), dict(B=37, S=10, c=1.0, t=1e-4)],
             lr=1e-3, Ws_lr=1e-4, lam=1, prior_rows="1")
  # C3 MEG-like: ragged wide groups, H=1, K=10, bigger batch.
  run_config("c3_meg_like",
             dict(D=256, K=10, H=1, M=0, gen_type="linear", group_dims=[37, 64, 5, 130, 1, 19]), seed=6,
             steps=[dict(B=256, S=1, c=0.1, t=1e-6)] * 3,
             lr=1e-3, Ws_lr=1e-6, lam=1, prior_rows="1", data="randn")
  # C4-like tiny: two-layer generators at MFMA-friendly sizes.
  run_config("c4_tiny32",
             dict(D=128, K=32, H=32, M=32, gen_type="mlp", group_dims=[32] * 4), seed=7,
             steps=[dict(B=64, S=1, c=1.0, t=1e-4)] * 3,
             lr=1e-3, Ws_lr=1e-4, lam=1, prior_rows="1")
  run_config("c4_tiny64",
             dict(D=128, K=64, H=64, M=64, gen_type="mlp", group_dims=[64] * 2), seed=8,
             steps=[dict(B=160, S=1, c=1.0, t=1e-4)] * 2,
             lr=1e-3, Ws_lr=1e-4, lam=1, prior_rows="1")
  # two-layer shapes with 32-wide latent / group input / hidden layer and 64 outputs per group (run through the all-64
  # kernel on zero-padded copies, csrc/dec_pad64.hip): an odd number of groups, ragged rows
  run_config("c4_pad32",
             dict(D=192, K=32, H=32, M=32, gen_type="mlp", group_dims=[64] * 3), seed=31,
             steps=[dict(B=96, S=1, c=1.0, t=1e-4)] * 2 + [dict(B=45, S=2, c=0.5, t=1e-4)],
             lr=1e-3, Ws_lr=1e-4, lam=1, prior_rows="1")
  run_config("c4_pad_m32",
             dict(D=128, K=64, H=64, M=32, gen_type="mlp", group_dims=[64] * 2), seed=32,
             steps=[dict(B=64, S=1, c=1.0, t=1e-4)] * 2,
             lr=1e-3, Ws_lr=1e-4, lam=1, prior_rows="1")
  # 32 outputs per group under 64-wide layers (second layer padded to 64 outputs, X read by halves): odd G, ragged rows
  run_config("c4_pad_dg32",
             dict(D=160, K=64, H=64, M=64, gen_type="mlp", group_dims=[32] * 5), seed=33,
             steps=[dict(B=80, S=1, c=1.0, t=1e-4)] * 2 + [dict(B=33, S=2, c=0.5, t=1e-4)],
             lr=1e-3, Ws_lr=1e-4, lam=1, prior_rows="1")
  # odd two-layer shapes, S=2
  run_config("c4_mixed",
             dict(D=24, K=16, H=8, M=24, gen_type="mlp", group_dims=[5, 12, 7]), seed=9,
             steps=[dict(B=48, S=2, c=0.5, t=1e-3)] * 3,
             lr=1e-3, Ws_lr=1e-3, lam=1, prior_rows="1")
  # C5-like: many tiny groups, strong threshold (about half of the rows die).
  run_config("c5_wide",
             dict(D=256, K=32, H=8, M=0, gen_type="linear", group_dims=[4] * 64), seed=10,
             steps=[dict(B=16, S=1, c=1.0, t=2.7)] + [dict(B=16, S=1, c=1.0, t=0.3)] * 2,
             lr=1e-3, Ws_lr=1e-4, lam=1, prior_rows="1")
  # All-64 two-layer shape (the fp16-term kernels' shape) driven into the states training produces: 30 steps with a
  # threshold that kills rows all along (and a weak one at the end under which some revive), one group that is dead
  # from the first step, weak rows scattered over a healthy group, lr three times the scale-up config's, the encoder's
  # std head switched 0.1 -> 1 on the way, a ragged two-sample batch at the end.  (Ws_lr is small against t so that the
  # threshold, not the gradient step, decides which rows live -- with Ws_lr = 1e-4 and this loss scale no row dies in
  # 30 steps; and lr = 2e-2 makes the RUN diverge, every trajectory within 1e-7 of it included.)
  def sparse_start(inference_net, gen):
    with torch.no_grad():
      gen.Ws[1] *= 0.18            # dies within the first steps
      gen.Ws[3] *= 1e-3            # a1 of this group is tiny beside its partner's in the workgroup
      gen.Ws[3, :40] = 0.0         # and most of its rows are already exactly zero
      gen.Ws[0, ::3] *= 0.3        # scattered weak rows in a healthy group
  run_config("c4_trained64",
             dict(D=320, K=64, H=64, M=64, gen_type="mlp", group_dims=[64] * 5), seed=21,
             steps=[dict(B=64, S=1, c=0.1, t=0.28)] * 10 + [dict(B=64, S=1, c=1.0, t=0.28)] * 16 +
                   [dict(B=45, S=2, c=1.0, t=0.05)] * 4,
             lr=3e-3, Ws_lr=1e-6, lam=1, prior_rows="1", full_steps={0, 11, 12, 28, 29}, prepare=sparse_start,
             state_steps={11, 28})
  if not ONLY or "prox_vectors" in ONLY:
    prox_vectors()
  if not ONLY or "prox_vectors_wide" in ONLY:
    prox_vectors_wide()


if __name__ == "__main__":
  main()
