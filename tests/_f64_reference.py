"""An independent float64 statement of the step's forward and backward on the GPU, for sizes no CPU oracle reaches
(the BASELINE scale-up shape: G = 1024, B = 65536).

Test infrastructure, not product code: plain torch float64 tensors, `einsum` over all groups at once and autograd
-- no kernel of the library, no fp16/bf16 anywhere -- of

    lib/models.py:91-95          h_g = z Ws[g];  mu_g = mu_net_g(h_g);  sigma_g = exp(log_sigma_g) + 1e-3
    lib/distributions.py:40-45   log N(x; mu, sigma), one global sum
    lib/utils.py:9-18            KL(q || prior)
    lib/oivae.py:15-23,50-71     weight prior; z_kl / B, loglik / S / B, penalty, loss, elbo

The batch is walked in row chunks (every term but the weight prior is a sum over rows, so the chunks' partial losses
add and autograd accumulates their gradients); X stays fp32 in HBM and is widened one chunk at a time.
"""
import math

import torch

LOG2PI = float(torch.log(torch.tensor(2 * math.pi, dtype=torch.float32)))   # the reference's fp32 constant, lib/distributions.py:9


def f64_step(eng, X, eps, c, sig_off=1e-3, gen_sig_off=1e-3, sigma_theta=1.0, lam=1.0, rows_per_chunk=2048,
             prior_mu=None, prior_sigma=None):
  """Six scalars (python floats, float64 arithmetic) and the gradient of `loss` w.r.t. every parameter, as a dict of
  float64 tensors in the engine's segment names and shapes.  Reads the engine's parameter VALUES only."""
  s = eng.spec
  dev = X.device
  P = {k: v.detach().double().clone().requires_grad_(True) for k, v in eng.p.items() if v.numel()}
  B, S = X.size(0), eps.size(0)
  G, K, H, M = s.G, s.K, s.H, s.M
  offs = [int(o) for o in eng.group_off_host]
  uniform = len(set(s.group_dims)) == 1
  dg = s.group_dims[0]
  pm = torch.zeros(1, K, dtype=torch.float64, device=dev) if prior_mu is None else prior_mu.double().view(-1, K)
  ps = torch.ones(1, K, dtype=torch.float64, device=dev) if prior_sigma is None else prior_sigma.double().view(-1, K)
  kl_sum = 0.0
  ll_sum = 0.0
  for lo in range(0, B, rows_per_chunk):
    hi = min(B, lo + rows_per_chunk)
    Xc = X[lo:hi].double()
    mu = Xc @ P["enc_Wmu"].t() + P["enc_bmu"]
    sig = torch.exp(Xc @ P["enc_Wsig"].t() + P["enc_bsig"]) * c + sig_off
    pmc = pm if pm.size(0) == 1 else pm[lo:hi]
    psc = ps if ps.size(0) == 1 else ps[lo:hi]
    kl = torch.sum(-0.5 + (sig ** 2 + (mu - pmc) ** 2) / (2 * psc ** 2) + torch.log(psc) - torch.log(sig))
    ll = 0.0
    for smp in range(S):
      z = mu + sig * eps[smp, lo:hi].double()
      if uniform:
        h = torch.einsum("rk,gkh->grh", z, P["Ws"])
        if s.gen_type == "mlp":
          h = torch.tanh(torch.einsum("grh,gmh->grm", h, P["gen_W1"]) + P["gen_b1"][:, None, :])
        elif s.gen_type == "tanh_linear":
          h = torch.tanh(h)
        W2 = P["gen_W2"].view(G, dg, -1)
        mux = torch.einsum("grm,gdm->rgd", h, W2).reshape(hi - lo, G * dg) + P["gen_b2"]
      else:
        cols = []
        for g in range(G):
          h = z @ P["Ws"][g]
          if s.gen_type == "mlp":
            h = torch.tanh(h @ P["gen_W1"][g].t() + P["gen_b1"][g])
          elif s.gen_type == "tanh_linear":
            h = torch.tanh(h)
          cols.append(h @ P["gen_W2"][offs[g]:offs[g + 1]].t() + P["gen_b2"][offs[g]:offs[g + 1]])
        mux = torch.cat(cols, dim=1)
      sx = torch.exp(P["log_sigma"]) + gen_sig_off
      ll = ll + torch.sum(-0.5 * LOG2PI - torch.log(sx) - 0.5 * ((Xc - mux) / sx) ** 2)
      del mux, h
    part = kl / B - ll / S / B
    part.backward()
    kl_sum += float(kl.detach())
    ll_sum += float(ll.detach())
    del Xc, mu, sig, kl, ll, part
  n_theta = 0
  th_sq = 0.0
  for k in ("gen_W1", "gen_b1", "gen_W2", "gen_b2"):
    if k in P:
      n_theta += P[k].numel()
      th_sq = th_sq + torch.sum(P[k] ** 2)
  lp_theta = n_theta * (-0.5 * LOG2PI - math.log(sigma_theta)) - 0.5 * th_sq / sigma_theta ** 2
  (-lp_theta).backward()
  penalty = float(torch.sum(torch.sqrt(torch.sum(P["Ws"].detach() ** 2, dim=2))))
  z_kl, loglik, lpt = kl_sum / B, ll_sum / S / B, float(lp_theta.detach())
  loss = -(-z_kl + loglik + lpt)
  lpW = -lam * penalty
  scalars = dict(z_kl=z_kl, loglik_term=loglik, logprob_theta=lpt, logprob_W=lpW, loss=loss, elbo=-loss + lpW)
  grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in P.items()}
  return scalars, grads
