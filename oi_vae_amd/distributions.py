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


class Gamma(object):
  """Gamma(shape, rate) observation model (lib/distributions.py:59-71).  The reference's
  class has only `logprob`; `size` is added so that `_DistributionCat.logprob` can split
  its argument (in the reference that call raises AttributeError for a Gamma member)."""

  def __init__(self, shape, rate):
    assert shape.size() == rate.size()
    self.shape = shape
    self.rate = rate

  def size(self, *args, **kwargs):
    return self.shape.size(*args, **kwargs)

  def logprob(self, x):
    return torch.sum(self.shape * torch.log(self.rate) - torch.lgamma(self.shape)
                     + (self.shape - 1) * torch.log(x) - self.rate * x)


class _DistributionCat(object):
  """Concatenation of arbitrary member distributions along `dim`
  (lib/distributions.py:73-86): evaluation-side only, the training step's fused
  kernels take Normal outputs."""

  def __init__(self, distributions, dim=-1):
    self.distributions = distributions
    self.dim = dim

  def sample(self):
    return torch.cat([d.sample() for d in self.distributions], dim=self.dim)

  def logprob(self, x):
    xs = torch.split(x, [d.size(self.dim) for d in self.distributions], dim=self.dim)
    return sum(d.logprob(xx) for d, xx in zip(self.distributions, xs))

  def entropy(self):
    return sum(d.entropy() for d in self.distributions)


def DistributionCat(distributions, dim=-1):
  if all(isinstance(d, Normal) for d in distributions):
    return Normal(mu=torch.cat([d.mu for d in distributions], dim=dim),
                  sigma=torch.cat([d.sigma for d in distributions], dim=dim))
  return _DistributionCat(distributions, dim=-1)
