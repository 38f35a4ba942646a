"""`NormalNet` and `BayesianGroupLassoGenerator` with the reference's interface
(/root/reference/lib/models.py:10-26,73-119).

The objects are built exactly as the reference's scripts build them (torch
modules for the nets, a leaf tensor `Ws`).  `OIVAE` recognises their structure
and moves their storage into the engine's flat device buffers; after that these
tensors are views of engine memory and the training step runs in HIP kernels.
"""
import ctypes
import itertools

import torch

from . import _lib
from .distributions import DistributionCat, Normal


class NormalNet(object):
  def __init__(self, mu_net, sigma_net):
    self.mu_net = mu_net
    self.sigma_net = sigma_net

  def __call__(self, x):
    return Normal(self.mu_net(x), self.sigma_net(x))

  def parameters(self):
    return itertools.chain(self.mu_net.parameters(), self.sigma_net.parameters())

  def cuda(self):
    self.mu_net.cuda()
    self.sigma_net.cuda()


class BayesianGroupLassoGenerator(object):
  """Group-lasso sparse first layer `Ws` [G, K, H] followed by one generator
  net per group (lib/models.py:73-119)."""

  def __init__(self, group_generators, group_input_dim, dim_z):
    self.group_generators = group_generators
    self.group_input_dim = group_input_dim
    self.dim_z = dim_z
    self.num_groups = len(group_generators)
    self.Ws = torch.randn(self.num_groups, self.dim_z, self.group_input_dim).requires_grad_(True)
    self._engine = None  # set by OIVAE when the model is adopted

  def __call__(self, z):
    # inference-time path (plots, reconstructions); the training step never calls it
    return DistributionCat([gen(z @ self.Ws[i]) for i, gen in enumerate(self.group_generators)], dim=-1)

  def group_generators_parameters(self):
    return itertools.chain(*[gen.parameters() for gen in self.group_generators])

  def parameters(self):
    return itertools.chain(*[gen.parameters() for gen in self.group_generators], [self.Ws])

  def proximal_step(self, t):
    """Block soft-threshold of every row Ws[g,k,:] in place (lib/models.py:106-111),
    by the HIP kernel; bit-identical to the reference's CPU arithmetic."""
    if self._engine is not None:
      self._engine.proximal_step(t)
      return
    W = self.Ws.data
    if not W.is_cuda:
      raise _lib.OivaeError("proximal_step runs on the GPU: call .cuda() on the model or hand it to OIVAE first")
    if not W.is_contiguous():
      raise _lib.OivaeError("Ws must be contiguous")
    L = _lib.init_device(W.device.index)
    with torch.cuda.device(W.device):
      st = ctypes.c_void_p(torch.cuda.current_stream(W.device).cuda_stream)
      _lib.check(L.oivae_prox_f32(W.data_ptr(), W.size(0) * W.size(1), W.size(2), float(t), None, None, st),
                 "oivae_prox_f32")

  def group_lasso_penalty(self):
    return torch.sum(torch.sqrt(torch.sum(torch.pow(self.Ws, 2), dim=2)))

  def connectivity(self):
    """[G, K] matrix of row norms ||Ws[g,k,:]|| -- the paper's latent-to-group
    connectivity figure (bars_data_oivae.py:200-210)."""
    return torch.sqrt(torch.sum(torch.pow(self.Ws.data, 2), dim=2))

  def cuda(self):
    # like the reference, this REBINDS Ws: build optimizers after the move
    self.Ws = self.Ws.data.cuda().requires_grad_(True)
    for gen in self.group_generators:
      gen.cuda()
