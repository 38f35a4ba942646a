"""Diagonal Gaussian, Gamma and distribution concatenation with the reference's
interface (/root/reference/lib/distributions.py:27-95).

These objects serve the calls the scripts make outside the training step
(`generative_net(inference_net(X).mu).sample()` for plots,
bars_data_oivae.py:133-149).  Inside `OIVAE.step` the same quantities are
computed by the HIP kernels and only wrapped in a `Normal` for the return dict.
"""
import math

import torch

LOG2PI = torch.log(torch.tensor([2 * math.pi], dtype=torch.float32))[0]


class Normal(object):
  def __init__(self, mu, sigma):
    assert mu.size() == sigma.size()
    self.mu = mu
    self.sigma = sigma

  def size(self, *args, **kwargs):
    return self.mu.size(*args, **kwargs)

  def sample(self):
    # like the reference, eps comes from the CPU default generator, so a run
    # seeded with torch.manual_seed draws the same numbers on either code base
    eps = torch.randn(self.mu.size()).to(device=self.mu.device, dtype=self.mu.dtype)
    return self.mu + self.sigma * eps

  def logprob(self, x):
    lp = LOG2PI.to(self.mu.device)
    return torch.sum(-0.5 * lp - torch.log(self.sigma) - 0.5 * torch.pow((x - self.mu) / self.sigma, 2))

  def expand(self, *args, **kwargs):
    return Normal(self.mu.expand(*args, **kwargs), self.sigma.expand(*args, **kwargs))

  def expand_as(self, other):
    # the reference passes a Normal here (lib/oivae.py:50); take its size
    size = other.size()
    return Normal(self.mu.expand(size), self.sigma.expand(size))


class _GammaLogProb(torch.autograd.Function):
  """Gamma.logprob on device tensors through oivae_gamma_logprob_f32: the log-likelihood and the gradients autograd
  would derive from the reference's expression (lib/distributions.py:65-71), inputs read once.  No CPU route: host
  tensors take the torch expression in Gamma.logprob."""

  @staticmethod
  def forward(ctx, shape, rate, x):
    import ctypes

    from . import _lib
    L = _lib.init_device(shape.device.index)
    a, b, v = shape.contiguous(), rate.contiguous(), x.contiguous()
    need = [ctx.needs_input_grad[i] for i in range(3)]
    grads = [torch.empty_like(a) if n else None for n in need]
    out = torch.empty(1, device=a.device, dtype=torch.float32)
    ws_bytes = int(L.oivae_gamma_workspace_bytes())
    ws = torch.empty(ws_bytes // 8, device=a.device, dtype=torch.float64)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    with torch.cuda.device(a.device):
      rc = L.oivae_gamma_logprob_f32(ptr(a), ptr(b), ptr(v), a.numel(), ptr(out), ptr(grads[0]), ptr(grads[1]), ptr(grads[2]),
                                     ptr(ws), ws_bytes, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "oivae_gamma_logprob_f32")
    ctx.grads = grads
    return out[0]

  @staticmethod
  def backward(ctx, g):
    return tuple(None if d is None else d * g for d in ctx.grads)


class Gamma(object):
  """Gamma(shape, rate) observation model for non-negative data (counterpart of
  lib/distributions.py:59-71, which has `logprob` only).  `size` and `sample` are additions: without
  `size` a Gamma member cannot be split off by `_DistributionCat.logprob` (in the reference that call
  raises AttributeError)."""

  def __init__(self, shape, rate):
    if shape.shape != rate.shape:
      raise AssertionError("Gamma: shape and rate tensors differ in size")
    self.shape = shape
    self.rate = rate

  def size(self, *args, **kwargs):
    return self.shape.size(*args, **kwargs)

  def _dist(self):
    return torch.distributions.Gamma(self.shape, self.rate, validate_args=False)

  def sample(self):
    return self._dist().sample()

  def logprob(self, x):
    """Sum over all elements of log Gamma(x; shape, rate) = a log b - lgamma(a) + (a - 1) log x - b x, the four terms
    added in that order (lib/distributions.py:65-70): the same bits as the reference, and its NaN at x = 0 with
    shape 1 (0 * log 0), which torch.distributions' xlogy would turn into 0."""
    a, b = self.shape, self.rate
    if x.is_cuda and a.is_cuda and b.is_cuda and a.dtype == b.dtype == x.dtype == torch.float32 and a.shape == x.shape:
      return _GammaLogProb.apply(a, b, x)   # one pass over the data on the GPU (csrc/gamma.hip), forward and gradients
    return torch.sum(a * torch.log(b) - torch.lgamma(a) + (a - 1) * torch.log(x) - b * x)


class _DistributionCat(object):
  """Independent member distributions side by side along `dim` (counterpart of
  lib/distributions.py:73-86): a sample is the members' samples concatenated, the log-probability of a
  concatenated x the sum of the members' over their own slices.  Evaluation side only: the training
  step's kernels take Normal outputs (DistributionCat below folds those into one Normal)."""

  def __init__(self, distributions, dim=-1):
    self.distributions = list(distributions)
    self.dim = dim

  def _widths(self):
    return [d.size(self.dim) for d in self.distributions]

  def sample(self):
    return torch.cat([d.sample() for d in self.distributions], dim=self.dim)

  def logprob(self, x):
    total = 0
    for d, piece in zip(self.distributions, torch.split(x, self._widths(), dim=self.dim)):
      total = total + d.logprob(piece)
    return total

  def entropy(self):
    total = 0
    for d in self.distributions:
      total = total + d.entropy()
    return total


def DistributionCat(distributions, dim=-1):
  if all(isinstance(d, Normal) for d in distributions):
    return Normal(mu=torch.cat([d.mu for d in distributions], dim=dim),
                  sigma=torch.cat([d.sigma for d in distributions], dim=dim))
  return _DistributionCat(distributions, dim=-1)
