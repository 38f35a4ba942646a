"""CPU restatement of the oi-VAE training step (TEST INFRASTRUCTURE ONLY).

This is the checker for the HIP engine, never the thing shipped or measured
(except as the ``cpu_baseline`` leg of ``bench.py``).  It restates, with its own
data model (plain dict of arrays, per-group Python loop, torch CPU autograd),
the algorithm of the reference's hot path:

  * training step            /root/reference/lib/oivae.py:42-90
  * weight prior             /root/reference/lib/oivae.py:15-23
  * grouped decoder          /root/reference/lib/models.py:91-95
  * group-lasso penalty      /root/reference/lib/models.py:113-114
  * proximal step            /root/reference/lib/models.py:106-111
  * diagonal Gaussian        /root/reference/lib/distributions.py:36-45
  * KL of Gaussians          /root/reference/lib/utils.py:9-18
  * encoder / generator heads as wired by the scripts
                             /root/reference/bars_data_oivae.py:66-108
                             /root/reference/mo_mo_mocap_oivae.py:195-212

Pinning: ``tests/golden/*.npz`` were produced by importing the real reference
from ``/root/reference`` in the build container (script
``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` checks this
restatement against them.  The arithmetic of the reference lives in PyTorch's
CPU kernels (torch 2.10.0), so every float result here is tolerance-checked,
EXCEPT the proximal step, whose zero set has to be bit-exact: that part is
restated in C with every rounding spelled out (``oracle/prox_oracle.c``: the
row-sum order of torch's CPU kernel and the MKL-VML square root torch.sqrt
resolves to) so that it gives the same bits on any host;
``row_sumsq_reference_order`` below is a second, numpy statement of the sum
order used to cross-check the C one.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

# /root/reference/lib/distributions.py:9 -- log of the fp32-rounded 2*pi, in fp32.
LOG2PI = torch.log(torch.tensor([2 * math.pi], dtype=torch.float32))[0]

GEN_TYPES = ("linear", "tanh_linear", "mlp")


# --------------------------------------------------------------------------
# proximal step, bit-exact restatement
# --------------------------------------------------------------------------
def row_sumsq_reference_order(w: np.ndarray) -> np.ndarray:
  """sum(pow(w, 2), axis=-1) in the order torch 2.10's CPU kernel uses for a
  contiguous fp32 inner dimension (/root/reference/lib/models.py:107-109 calls
  torch.sum(torch.pow(Ws, 2), dim=2)).

  Order (probed in the build container and pinned by tests/golden/prox_*.npz):
    H < 8 : four interleaved scalar partial sums p[k] = sum_i a[4i+k]; the
            H%4 leftovers are added to p[0]; then p[0]+p[1]+p[2]+p[3] left to
            right.
    H >= 8: vectors of 8 lanes, V = H//8 of them; four interleaved vector
            partial sums P[k] = sum_i vec[4i+k] over the first 4*(V//4)
            vectors, the V%4 remaining vectors are added to P[0], then
            P[0]+P[1]+P[2]+P[3] left to right; the scalar result starts at 0,
            takes the H%8 tail elements left to right and then lanes 0..7 of
            the combined vector left to right.
  torch cascades once a partial sum takes 16 or more terms (H >= 512, aten
  SumKernel multi_row_sum): every 16 steps of four vectors the four level-0
  partial sums are added into level-1 ones and cleared, every 256 steps level 1
  into level 2, every 4096 level 2 into level 3; the steps behind the last
  full block of 16 go to level 0, then levels 1..3 are added to level 0 in
  that order and the sum goes on as above.
  Every product and every sum is rounded to fp32 (no FMA).
  """
  w = np.asarray(w, dtype=np.float32)
  H = w.shape[-1]
  sq = (w * w).astype(np.float32)
  lead = sq.shape[:-1]
  if H < 8:
    nq = H // 4
    p = [np.zeros(lead, np.float32) for _ in range(4)]
    for i in range(nq):
      for k in range(4):
        p[k] = (p[k] + sq[..., 4 * i + k]).astype(np.float32)
    for i in range(nq * 4, H):
      p[0] = (p[0] + sq[..., i]).astype(np.float32)
    for k in range(1, 4):
      p[0] = (p[0] + p[k]).astype(np.float32)
    return p[0]
  V = H // 8
  vec = sq[..., : V * 8].reshape(lead + (V, 8))
  P = [np.zeros(lead + (8,), np.float32) for _ in range(4)]
  nq = V // 4
  levels = [[np.zeros(lead + (8,), np.float32) for _ in range(4)] for _ in range(3)]   # levels 1..3
  i = 0
  while i + 16 <= nq:
    for _ in range(16):
      for k in range(4):
        P[k] = (P[k] + vec[..., 4 * i + k, :]).astype(np.float32)
      i += 1
    below = P
    for lev in range(3):
      for k in range(4):
        levels[lev][k] = (levels[lev][k] + below[k]).astype(np.float32)
        below[k] = np.zeros(lead + (8,), np.float32)
      if i & (15 << (4 * (lev + 1))):
        break
      below = levels[lev]
  for i in range(i, nq):
    for k in range(4):
      P[k] = (P[k] + vec[..., 4 * i + k, :]).astype(np.float32)
  for lev in range(3):
    for k in range(4):
      P[k] = (P[k] + levels[lev][k]).astype(np.float32)
  for i in range(nq * 4, V):
    P[0] = (P[0] + vec[..., i, :]).astype(np.float32)
  for k in range(1, 4):
    P[0] = (P[0] + P[k]).astype(np.float32)
  acc = np.zeros(lead, np.float32)
  for i in range(V * 8, H):
    acc = (acc + sq[..., i]).astype(np.float32)
  for lane in range(8):
    acc = (acc + P[0][..., lane]).astype(np.float32)
  return acc


def proximal_step_reference_order(Ws: np.ndarray, t: float):
  """Block soft-threshold of every row Ws[g,k,:]
  (/root/reference/lib/models.py:106-111): n = sqrt(sum w^2);
  w = (w / max(n, 1e-16)) * max(n - t, 0), two roundings per element.
  The square root is NOT the correctly rounded one: the reference's torch.sqrt
  resolves to MKL VML vmsSqrt(HA), restated in oracle/prox_oracle.c, which this
  calls.  Returns (new Ws, zero mask [G,K] uint8 = row is all zeros afterwards)."""
  from . import prox_oracle
  out, mask, _ = prox_oracle.proximal_step(Ws, t)
  return out, mask


# --------------------------------------------------------------------------
# model state
# --------------------------------------------------------------------------
def init_params(spec: Dict, seed: int = 0) -> Dict[str, object]:
  """Random initial parameters with the reference's initialisers: nn.Linear
  defaults for encoder and generators (bars_data_oivae.py:66-102), randn for
  Ws (lib/models.py:86-89), log(1e-2) for log_sigma (bars_data_oivae.py:92-95).
  Returned as numpy arrays / lists of numpy arrays (ragged groups)."""
  g = torch.Generator().manual_seed(seed)
  D, K, H = spec["D"], spec["K"], spec["H"]
  dims = list(spec["group_dims"])
  G = len(dims)
  assert sum(dims) == D
  typ = spec["gen_type"]
  M = spec.get("M", 0)

  def linear(out_f, in_f):
    bound = 1.0 / math.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound
    b = (torch.rand(out_f, generator=g) * 2 - 1) * bound
    return w.numpy().copy(), b.numpy().copy()

  p: Dict[str, object] = {}
  p["enc_Wmu"], p["enc_bmu"] = linear(K, D)
  p["enc_Wsig"], p["enc_bsig"] = linear(K, D)
  if typ == "mlp":
    w1b1 = [linear(M, H) for _ in range(G)]
    p["gen_W1"] = [w for w, _ in w1b1]
    p["gen_b1"] = [b for _, b in w1b1]
    w2b2 = [linear(d, M) for d in dims]
  else:
    w2b2 = [linear(d, H) for d in dims]
  p["gen_W2"] = [w for w, _ in w2b2]
  p["gen_b2"] = [b for _, b in w2b2]
  p["log_sigma"] = [np.full((d,), math.log(1e-2), np.float32) for d in dims]
  p["Ws"] = torch.randn(G, K, H, generator=g).numpy().copy()
  return p


class OracleModel:
  """Leaf tensors of one oi-VAE, one tensor per group as in the reference."""

  def __init__(self, spec: Dict, params: Dict[str, object]):
    self.spec = dict(spec)
    self.dims = list(spec["group_dims"])
    self.G = len(self.dims)
    self.typ = spec["gen_type"]
    assert self.typ in GEN_TYPES
    t = lambda a: torch.tensor(np.asarray(a, np.float32), requires_grad=True)
    self.enc_Wmu, self.enc_bmu = t(params["enc_Wmu"]), t(params["enc_bmu"])
    self.enc_Wsig, self.enc_bsig = t(params["enc_Wsig"]), t(params["enc_bsig"])
    self.gen_W1 = [t(a) for a in params.get("gen_W1", [])]
    self.gen_b1 = [t(a) for a in params.get("gen_b1", [])]
    self.gen_W2 = [t(a) for a in params["gen_W2"]]
    self.gen_b2 = [t(a) for a in params["gen_b2"]]
    self.log_sigma = [t(a) for a in params["log_sigma"]]
    self.Ws = t(params["Ws"])

  # parameter sets as the scripts hand them to the optimizers
  # (bars_data_oivae.py:234-244)
  def encoder_params(self) -> List[torch.Tensor]:
    return [self.enc_Wmu, self.enc_bmu, self.enc_Wsig, self.enc_bsig]

  def generator_params(self) -> List[torch.Tensor]:
    out: List[torch.Tensor] = []
    for g in range(self.G):
      if self.typ == "mlp":
        out += [self.gen_W1[g], self.gen_b1[g]]
      out += [self.gen_W2[g], self.gen_b2[g]]
    return out

  def adam_params(self) -> List[torch.Tensor]:
    return self.encoder_params() + self.generator_params() + list(self.log_sigma)

  def export(self) -> Dict[str, object]:
    n = lambda x: x.detach().numpy().copy()
    out = dict(enc_Wmu=n(self.enc_Wmu), enc_bmu=n(self.enc_bmu),
               enc_Wsig=n(self.enc_Wsig), enc_bsig=n(self.enc_bsig),
               gen_W2=[n(a) for a in self.gen_W2], gen_b2=[n(a) for a in self.gen_b2],
               log_sigma=[n(a) for a in self.log_sigma], Ws=n(self.Ws))
    if self.typ == "mlp":
      out["gen_W1"] = [n(a) for a in self.gen_W1]
      out["gen_b1"] = [n(a) for a in self.gen_b1]
    return out

  def export_grads(self) -> Dict[str, object]:
    def gr(x):
      return x.grad.detach().numpy().copy() if x.grad is not None else np.zeros(tuple(x.shape), np.float32)
    out = dict(enc_Wmu=gr(self.enc_Wmu), enc_bmu=gr(self.enc_bmu),
               enc_Wsig=gr(self.enc_Wsig), enc_bsig=gr(self.enc_bsig),
               gen_W2=[gr(a) for a in self.gen_W2], gen_b2=[gr(a) for a in self.gen_b2],
               log_sigma=[gr(a) for a in self.log_sigma], Ws=gr(self.Ws))
    if self.typ == "mlp":
      out["gen_W1"] = [gr(a) for a in self.gen_W1]
      out["gen_b1"] = [gr(a) for a in self.gen_b1]
    return out

  # ---- forward pieces -----------------------------------------------------
  def encode(self, X: torch.Tensor, c: float, sig_offset: float = 1e-3):
    """bars_data_oivae.py:66-87: mu = Linear(X); sigma = exp(Linear(X))*c + 1e-3."""
    mu = torch.nn.functional.linear(X, self.enc_Wmu, self.enc_bmu)
    sig = torch.exp(torch.nn.functional.linear(X, self.enc_Wsig, self.enc_bsig)) * c + sig_offset
    return mu, sig

  def decode(self, z: torch.Tensor, sig_offset: float = 1e-3):
    """lib/models.py:91-95 + lib/distributions.py:88-93: per group
    gen_g(z @ Ws[g]) then concatenate mu and sigma on the last dim."""
    mus, sigs = [], []
    for g in range(self.G):
      h = z @ self.Ws[g]
      if self.typ == "linear":
        mu = torch.nn.functional.linear(h, self.gen_W2[g], self.gen_b2[g])
      elif self.typ == "tanh_linear":
        mu = torch.nn.functional.linear(torch.tanh(h), self.gen_W2[g], self.gen_b2[g])
      else:
        a1 = torch.nn.functional.linear(h, self.gen_W1[g], self.gen_b1[g])
        mu = torch.nn.functional.linear(torch.tanh(a1), self.gen_W2[g], self.gen_b2[g])
      sig = torch.exp(self.log_sigma[g].expand(z.size(0), -1)) + sig_offset
      mus.append(mu)
      sigs.append(sig)
    return torch.cat(mus, dim=-1), torch.cat(sigs, dim=-1)


def normal_logprob(x, mu, sigma):
  """lib/distributions.py:40-45 (one global sum)."""
  return torch.sum(-0.5 * LOG2PI - torch.log(sigma) - 0.5 * torch.pow((x - mu) / sigma, 2))


def kl_normals(mu0, sig0, mu1, sig1):
  """lib/utils.py:9-18."""
  s0, s1 = torch.pow(sig0, 2), torch.pow(sig1, 2)
  return torch.sum(-0.5 + (s0 + torch.pow(mu0 - mu1, 2)) / (2 * s1) + torch.log(sig1) - torch.log(sig0))


def theta_logprob(model: OracleModel, sigma_theta: float):
  """lib/oivae.py:15-23: N(0, sigma_theta) on every mu_net parameter of every
  group generator (weights and biases; not log_sigma, not Ws)."""
  total = 0
  for p in model.generator_params():
    total = total + normal_logprob(p, torch.zeros_like(p), sigma_theta * torch.ones_like(p))
  return total


def group_lasso_penalty(Ws):
  """lib/models.py:113-114."""
  return torch.sum(torch.sqrt(torch.sum(torch.pow(Ws, 2), dim=2)))


def forward_terms(model: OracleModel, X, eps, c, lam, sigma_theta,
                  prior_mu=None, prior_sigma=None):
  """Everything lib/oivae.py:43-71 computes before the optimizers run.
  eps: [S,B,K] standard-normal draws (the reference draws them inside
  Normal.sample, lib/distributions.py:36-38; here they are an input so both
  sides see the same numbers)."""
  B = X.size(0)
  S = eps.size(0)
  mu_q, sig_q = model.encode(X, c)
  if prior_mu is None:
    prior_mu = torch.zeros(1, mu_q.size(1))
    prior_sigma = torch.ones(1, mu_q.size(1))
  z_kl = kl_normals(mu_q, sig_q, prior_mu.expand_as(mu_q), prior_sigma.expand_as(mu_q)) / B
  z = torch.cat([mu_q + sig_q * eps[s] for s in range(S)], dim=0)
  Xrep = X.detach().repeat(S, 1)
  mu_x, sig_x = model.decode(z)
  loglik = normal_logprob(Xrep, mu_x, sig_x) / S / B
  lp_theta = theta_logprob(model, sigma_theta)
  lp_W = -lam * group_lasso_penalty(model.Ws)
  loss = -1.0 * (-z_kl + loglik + lp_theta)
  elbo = -loss + lp_W
  return dict(mu_q=mu_q, sig_q=sig_q, z=z, z_kl=z_kl, loglik_term=loglik,
              logprob_theta=lp_theta, logprob_W=lp_W, loss=loss, elbo=elbo)


class OracleTrainer:
  """Model + the two optimizers the scripts build (bars_data_oivae.py:233-244):
  Adam over encoder, generator nets and log_sigmas; plain SGD over Ws."""

  def __init__(self, spec, params, lr, Ws_lr, lam, sigma_theta=1.0,
               prior_mu=None, prior_sigma=None):
    self.model = OracleModel(spec, params)
    self.lam = lam
    self.sigma_theta = sigma_theta
    self.prior_mu = None if prior_mu is None else torch.as_tensor(prior_mu, dtype=torch.float32)
    self.prior_sigma = None if prior_sigma is None else torch.as_tensor(prior_sigma, dtype=torch.float32)
    self.adam = torch.optim.Adam(self.model.adam_params(), lr=lr)
    self.sgd = torch.optim.SGD([self.model.Ws], lr=Ws_lr, momentum=0)

  def load_adam_state(self, exp_avg, exp_avg_sq, steps_done):
    """Adam's moments and step count from a recorded state (dicts in `OracleModel.export`'s layout): the next
    `step` then continues that run (teacher-forced parity checks)."""
    m = self.model
    def items(d):
      out = [d["enc_Wmu"], d["enc_bmu"], d["enc_Wsig"], d["enc_bsig"]]
      for g in range(m.G):
        if m.typ == "mlp":
          out += [d["gen_W1"][g], d["gen_b1"][g]]
        out += [d["gen_W2"][g], d["gen_b2"][g]]
      return out + list(d["log_sigma"])
    for p, a, b in zip(m.adam_params(), items(exp_avg), items(exp_avg_sq)):
      self.adam.state[p] = {"step": torch.tensor(float(steps_done)),
                            "exp_avg": torch.tensor(np.asarray(a, np.float32)).reshape(p.shape).clone(),
                            "exp_avg_sq": torch.tensor(np.asarray(b, np.float32)).reshape(p.shape).clone()}

  def step(self, X, eps, c, prox_step_size, want_grads=True, want_pre_prox=False):
    """One training step, lib/oivae.py:42-90.  Returns python scalars of the
    six reported terms (pre-update values), optionally the gradients, and the
    zero-row mask of Ws after the proximal step."""
    X = torch.as_tensor(X, dtype=torch.float32)
    eps = torch.as_tensor(eps, dtype=torch.float32)
    terms = forward_terms(self.model, X, eps, c, self.lam, self.sigma_theta,
                          self.prior_mu, self.prior_sigma)
    self.adam.zero_grad()
    self.sgd.zero_grad()
    terms["loss"].backward()
    info = {k: float(terms[k].detach()) for k in
            ("z_kl", "loglik_term", "logprob_theta", "logprob_W", "loss", "elbo")}
    info["mu_q"] = terms["mu_q"].detach().numpy().copy()
    info["sig_q"] = terms["sig_q"].detach().numpy().copy()
    info["z"] = terms["z"].detach().numpy().copy()
    if want_grads:
      info["grads"] = self.model.export_grads()
    self.adam.step()
    self.sgd.step()
    if want_pre_prox:
      info["Ws_pre_prox"] = self.model.Ws.detach().numpy().copy()
    if self.lam > 0:
      new_Ws, mask = proximal_step_reference_order(self.model.Ws.detach().numpy(), prox_step_size)
      with torch.no_grad():
        self.model.Ws.copy_(torch.from_numpy(new_Ws))
      info["zero_mask"] = mask
    else:
      info["zero_mask"] = np.all(self.model.Ws.detach().numpy() == 0, axis=-1).astype(np.uint8)
    return info
