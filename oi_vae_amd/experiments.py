"""Wiring shared by the three experiment drivers in scripts/ -- the callers of the hot path.

Everything here is what the reference's scripts spell out inline
(/root/reference/bars_data_oivae.py:60-108,212-298, mo_mo_mocap_oivae.py:158-325,
meg_oi_vae.py:130-256): the encoder with its `exp(a) * stddev_multiple + 1e-3` head, one
generator per group with a learned constant log_sigma, Adam over encoder + generators +
log_sigmas, plain SGD over Ws, `OIVAE.step(X, Ws_lr * lam * lam_adjustment, mc_samples)` per
minibatch, `stddev_multiple` flipped from 0.1 to 1 after iteration 1000, and the latent-to-group
connectivity matrix as the artefact.  The scripts only choose data and hyper-parameters.
"""
import json
import os

import numpy as np
import torch

from .distributions import Normal
from .models import BayesianGroupLassoGenerator, NormalNet
from .oivae import OIVAE, NormalPriorTheta
from .utils import Lambda


class Experiment(object):
  """Model + optimizers + trainer for one configuration."""

  def __init__(self, dim_x, dim_z, group_dims, group_input_dim, gen_type="linear", hidden=0,
               lr=1e-3, Ws_lr=1e-4, lam=1.0, lam_adjustment=1.0, prior_theta_scale=1.0,
               prior_rows=1, native_optimizers=True, device=None, device_eps=False):
    assert sum(group_dims) == dim_x
    self.stddev_multiple = 0.1  # bars_data_oivae.py:60; read through the closure below on every step
    self.lam, self.lam_adjustment, self.Ws_lr = lam, lam_adjustment, Ws_lr
    self.inference_net = NormalNet(
      mu_net=torch.nn.Sequential(torch.nn.Linear(dim_x, dim_z)),
      sigma_net=torch.nn.Sequential(
        torch.nn.Linear(dim_x, dim_z),
        Lambda(torch.exp),
        Lambda(lambda x: x * self.stddev_multiple + 1e-3)))

    def make_group_generator(out_dim):
      # log_sigma is not a registered parameter: free of the ridge prior (bars_data_oivae.py:90-95)
      log_sigma = torch.log(1e-2 * torch.ones(out_dim)).requires_grad_(True)
      if gen_type == "linear":            # bars_data_oivae.py:97, meg_oi_vae.py:178
        mu_net = torch.nn.Linear(group_input_dim, out_dim)
      elif gen_type == "tanh_linear":     # mo_mo_mocap_oivae.py:203-207
        mu_net = torch.nn.Sequential(torch.nn.Tanh(), torch.nn.Linear(group_input_dim, out_dim))
      elif gen_type == "mlp":             # BASELINE.json "2-layer generators"
        mu_net = torch.nn.Sequential(torch.nn.Linear(group_input_dim, hidden), torch.nn.Tanh(),
                                     torch.nn.Linear(hidden, out_dim))
      else:
        raise ValueError(gen_type)
      return NormalNet(
        mu_net=mu_net,
        sigma_net=Lambda(lambda x, log_sigma: torch.exp(log_sigma.expand(x.size(0), -1)) + 1e-3,
                         extra_args=(log_sigma,)))

    self.generative_net = BayesianGroupLassoGenerator(
      group_generators=[make_group_generator(d) for d in group_dims],
      group_input_dim=group_input_dim, dim_z=dim_z)
    adam_groups = [
      {'params': self.inference_net.parameters(), 'lr': lr},
      {'params': self.generative_net.group_generators_parameters(), 'lr': lr},
      {'params': [gen.sigma_net.extra_args[0] for gen in self.generative_net.group_generators], 'lr': lr}]
    sgd_groups = [{'params': [self.generative_net.Ws], 'lr': Ws_lr, 'momentum': 0}]
    if native_optimizers:  # the engine's fused Adam and SGD+prox kernels
      from . import optim
      self.optimizer, self.optimizer_Ws = optim.Adam(adam_groups), optim.SGD(sgd_groups)
    else:                  # caller-owned torch.optim, exactly as the scripts build them
      self.optimizer, self.optimizer_Ws = torch.optim.Adam(adam_groups), torch.optim.SGD(sgd_groups)
    prior_z = Normal(torch.zeros(prior_rows, dim_z), torch.ones(prior_rows, dim_z))
    self.vae = OIVAE(
      inference_model=self.inference_net, generative_model=self.generative_net, prior_z=prior_z,
      prior_theta=NormalPriorTheta(prior_theta_scale), lam=lam,
      optimizers=[self.optimizer, self.optimizer_Ws], device=device, device_eps=device_eps)
    self.device = self.vae.engine.device
    self.elbo_per_iter = []
    self.iteration = 0

  def step(self, X, mc_samples=1):
    if self.iteration > 1000:   # bars_data_oivae.py:258-259
      self.stddev_multiple = 1
    info = self.vae.step(X=X, prox_step_size=self.Ws_lr * self.lam * self.lam_adjustment, mc_samples=mc_samples)
    self.iteration += 1
    return info

  def report(self, info, header):
    """The five numbers every script prints (bars_data_oivae.py:293-298); one device->host sync."""
    vals = torch.stack([info[k] for k in ('elbo', 'z_kl', 'loglik_term', 'logprob_theta', 'logprob_W')]).tolist()
    self.elbo_per_iter.append((self.iteration, vals[0]))
    print(header)
    print('  ELBO:', vals[0])
    print('    -KL(q(z) || p(z))', -vals[1])
    print('    loglik_term      ', vals[2])
    print('    log p(theta)     ', vals[3])
    print('    log p(W)         ', vals[4])
    return vals

  def connectivity(self, normalise_per_latent=False):
    """[groups, dim_z] norms ||Ws[g,k,:]|| (bars_data_oivae.py:200-210); divided by each latent's largest
    entry for the mocap figure (mo_mo_mocap_oivae.py:236-239)."""
    C = self.generative_net.connectivity()
    if normalise_per_latent:
      C = C / torch.clamp(torch.max(C, dim=0)[0], min=1e-30)
    return C.cpu().numpy()

  def save_artifacts(self, outdir, name, group_names=None, normalise_per_latent=False):
    """Connectivity matrix (+ zero pattern) as .npy/.json, ELBO curve as .json, and the reference's imshow
    figure as a PDF when matplotlib is importable."""
    os.makedirs(outdir, exist_ok=True)
    C = self.connectivity(normalise_per_latent)
    np.save(os.path.join(outdir, name + '_connectivity_matrix.npy'), C)
    with open(os.path.join(outdir, name + '_summary.json'), 'w') as f:
      json.dump({'iterations': self.iteration, 'zero_connections': int((C == 0).sum()), 'connections': int(C.size),
                 'elbo_per_iter': self.elbo_per_iter, 'group_names': group_names}, f)
    try:
      import matplotlib
      matplotlib.use('Agg')
      import matplotlib.pyplot as plt
    except Exception:
      return C
    fig, ax = plt.subplots()
    ax.imshow(C, aspect='equal')
    ax.set_xlabel('dimensions of z')
    ax.set_ylabel('group generative nets')
    ax.xaxis.tick_top()
    ax.xaxis.set_label_position('top')
    if group_names is not None and len(group_names) <= 64:
      ax.yaxis.set_ticks(np.arange(len(group_names)))
      ax.yaxis.set_ticklabels(group_names)
    fig.savefig(os.path.join(outdir, name + '_connectivity_matrix.pdf'), format='pdf')
    plt.close(fig)
    return C


def minibatches(X, batch_size, generator=None):
  """One shuffled epoch like `DataLoader(shuffle=True)` without drop_last: the final batch may be ragged
  (mo_mo_mocap_oivae.py:150-155)."""
  perm = torch.randperm(X.size(0), generator=generator)
  for i in range(0, X.size(0), batch_size):
    yield X[perm[i:i + batch_size]]
