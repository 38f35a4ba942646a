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
