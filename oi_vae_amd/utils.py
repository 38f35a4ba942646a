"""`Lambda` and `KL_Normals` under the reference's names (/root/reference/lib/utils.py:9-28).

`Lambda` is the one class of that file the engine depends on: the scripts express
the encoder's `exp(.) * stddev_multiple + 1e-3` head and each generator's constant
`log_sigma` head as `Lambda` modules (bars_data_oivae.py:84-85,98-101), and
`oi_vae_amd.adopt` recognises those heads by this type and probes their constants.
`KL_Normals` is kept for scripts that call it directly; the training step computes
the same sum inside `enc_fwd_epilogue` (csrc/generic.hip).
"""
import torch


class Lambda(torch.nn.Module):
  """A function as a module: `forward(x)` = `func(x, *extra_args, **extra_kwargs)`.
  `extra_args` stays a public attribute: the scripts reach a generator's `log_sigma`
  through `gen.sigma_net.extra_args[0]` (bars_data_oivae.py:238)."""

  def __init__(self, func, extra_args=(), extra_kwargs=None):
    super().__init__()
    self.func = func
    self.extra_args = tuple(extra_args)
    self.extra_kwargs = dict(extra_kwargs or {})

  def forward(self, x):
    return self.func(x, *self.extra_args, **self.extra_kwargs)


def KL_Normals(d0, d1):
  """Sum over all elements of KL(N(mu0, sigma0) || N(mu1, sigma1)) for two diagonal Gaussians of
  one shape (`.mu`, `.sigma`); lib/oivae.py:50 divides it by the batch size."""
  if d0.mu.shape != d1.mu.shape:
    raise AssertionError("KL_Normals: shapes differ: %s vs %s" % (tuple(d0.mu.shape), tuple(d1.mu.shape)))
  var0, var1 = d0.sigma * d0.sigma, d1.sigma * d1.sigma
  # term order as lib/utils.py:13-18 adds them (fp32 sums are order-sensitive), and the two logarithms kept apart:
  # log(s1 / s0) over- or underflows where log s1 - log s0 does not
  per_element = -0.5 + (var0 + (d0.mu - d1.mu) ** 2) / (2 * var1) + torch.log(d1.sigma) - torch.log(d0.sigma)
  return per_element.sum()
