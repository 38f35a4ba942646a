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
  """x -> Normal(mu_net(x), sigma_net(x)): the encoder and every group generator are one of these
  (lib/models.py:10-26).  Plain container: `OIVAE` reads `mu_net` / `sigma_net` to recognise the model."""

  def __init__(self, mu_net, sigma_net):
    self.mu_net, self.sigma_net = mu_net, sigma_net

  def __call__(self, x):
    return Normal(mu=self.mu_net(x), sigma=self.sigma_net(x))

  def parameters(self):
    for net in (self.mu_net, self.sigma_net):
      yield from net.parameters()

  def cuda(self):
    for net in (self.mu_net, self.sigma_net):
      net.cuda()


class FirstLayerSparseDecoder(object):
  """Group-lasso sparse first layer for plain autoencoders, with the reference's
  interface (lib/models.py:29-71; driver data_vs_test_loss_autoencoder.py:31-50,
  96-145).  Forward and backward of this variant are ordinary torch modules (the
  reference trains it with autograd); the proximal step -- the operation the two
  decoder classes share -- runs in the HIP kernel `oivae_prox_cols_f32`, bit-identical
  to the reference's CPU arithmetic.  On the first GPU proximal step the G map weights
  are packed into one [sum H_g, K] device buffer (each nn.Linear weight becomes a view of
  it, so optimizers keep their Parameter objects) and one launch serves all maps."""

  def __init__(self, group_generators, group_generators_input_dims, input_dim):
    assert len(group_generators) == len(group_generators_input_dims)
    self.group_generators = group_generators
    self.group_generators_input_dims = group_generators_input_dims
    self.input_dim = input_dim
    self.latent_to_group_maps = [
      torch.nn.Linear(self.input_dim, k, bias=False)
      for k in self.group_generators_input_dims
    ]
    self._flat = None
    self._group_off = None
    self.zero_mask = None
    self.col_norms = None

  def __call__(self, z):
    outs = [gen(to_group(z)) for to_group, gen in zip(self.latent_to_group_maps, self.group_generators)]
    return torch.cat(outs, dim=-1)

  def parameters(self):
    for gen in self.group_generators:
      yield from gen.parameters()
    for to_group in self.latent_to_group_maps:
      yield from to_group.parameters()

  def group_lasso_penalty(self):
    """Sum over maps and latent columns of the column's Euclidean norm."""
    return sum(torch.linalg.vector_norm(to_group.weight, dim=0).sum() for to_group in self.latent_to_group_maps)

  def cuda(self):
    for gen in self.group_generators:
      gen.cuda()
    for m in self.latent_to_group_maps:
      m.cuda()
    self._flat = None
    self._packed()   # pack now: an optimizer built after .cuda() then holds the views the proximal kernel updates
    return self

  def _packed(self):
    maps = self.latent_to_group_maps
    dev = maps[0].weight.device
    if dev.type != "cuda":
      raise _lib.OivaeError("proximal_step runs on the GPU: call .cuda() on the decoder first (there is no CPU fallback)")
    K = self.input_dim
    ok = self._flat is not None and self._flat.device == dev
    if ok:
      off = 0
      for m in maps:  # still views of the packed buffer?
        h = m.weight.shape[0]
        ok = ok and m.weight.data_ptr() == self._flat.data_ptr() + off * K * 4 and m.weight.is_contiguous()
        off += h
    if not ok:
      dims = [int(m.weight.shape[0]) for m in maps]
      flat = torch.empty(sum(dims), K, dtype=torch.float32, device=dev)
      off = 0
      for m, h in zip(maps, dims):
        flat[off:off + h].copy_(m.weight.data)
        m.weight.data = flat[off:off + h]
        off += h
      self._flat = flat
      self._dims = dims
      self._group_off = torch.tensor([0] + list(itertools.accumulate(dims)), dtype=torch.int32, device=dev)
      self.zero_mask = torch.zeros(len(dims), K, dtype=torch.uint8, device=dev)
      self.col_norms = torch.zeros(len(dims), K, dtype=torch.float32, device=dev)
    return self._flat

  def proximal_step(self, t):
    """Block soft-threshold of every column of every map in place (lib/models.py:65-71)."""
    flat = self._packed()
    dev = flat.device
    L = _lib.init_device(dev.index)
    with torch.cuda.device(dev):
      st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
      _lib.check(L.oivae_prox_cols_f32(flat.data_ptr(), self._group_off.data_ptr(), len(self._dims), self.input_dim,
                                       max(self._dims), float(t), self.zero_mask.data_ptr(),
                                       self.col_norms.data_ptr(), st), "oivae_prox_cols_f32")

  def connectivity(self):
    """[G, K] matrix of column norms ||W_g[:, k]|| (latent k -> group g)."""
    return torch.stack([torch.sqrt(torch.sum(torch.pow(m.weight.data, 2), dim=0)) for m in self.latent_to_group_maps])


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
    # inference-time path (plots, reconstructions); the training step never calls it.  All groups' first maps
    # in one batched product: h[g] = z Ws[g]
    h = torch.einsum("nk,gkh->gnh", z, self.Ws)
    return DistributionCat([gen(h[g]) for g, gen in enumerate(self.group_generators)], dim=-1)

  def group_generators_parameters(self):
    for gen in self.group_generators:
      yield from gen.parameters()

  def parameters(self):
    yield from self.group_generators_parameters()
    yield self.Ws

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
    """Sum over groups and latent rows of ||Ws[g,k,:]|| (evaluated by `oivae_finalize_f32` inside the step)."""
    return torch.linalg.vector_norm(self.Ws, dim=2).sum()

  def connectivity(self):
    """[G, K] matrix of row norms ||Ws[g,k,:]|| -- the paper's latent-to-group
    connectivity figure (bars_data_oivae.py:200-210)."""
    return torch.sqrt(torch.sum(torch.pow(self.Ws.data, 2), dim=2))

  def cuda(self):
    # like the reference, this REBINDS Ws: build optimizers after the move
    self.Ws = self.Ws.data.cuda().requires_grad_(True)
    for gen in self.group_generators:
      gen.cuda()
