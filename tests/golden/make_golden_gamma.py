#!/usr/bin/env python3
"""Gamma log-likelihoods and their autograd gradients computed by the REAL reference's class
(/root/reference/lib/distributions.py:59-71) on seeded inputs in four regimes:
    python tests/golden/make_golden_gamma.py        (build container only)
Numbers only; nothing of the reference's source is written anywhere."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from lib.distributions import Gamma  # noqa: E402

g = torch.Generator().manual_seed(23)
out = {}


def case(name, shape, rate, x):
  a, b, v = (t.clone().requires_grad_(True) for t in (shape, rate, x))
  lp = Gamma(a, b).logprob(v)
  if bool(torch.isfinite(lp)):
    lp.backward()
    out.update({name + "/d_shape": a.grad.numpy(), name + "/d_rate": b.grad.numpy(), name + "/d_x": v.grad.numpy()})
  out.update({name + "/shape": shape.numpy(), name + "/rate": rate.numpy(), name + "/x": x.numpy(),
              name + "/logprob": np.float32(lp.item())})


def u(lo, hi, *size):
  return torch.rand(*size, generator=g) * (hi - lo) + lo


case("typical", u(0.2, 4.2, 151, 129), u(0.1, 3.1, 151, 129), u(1e-3, 5.0, 151, 129))
case("small_shape", u(1e-3, 0.1, 64, 33), u(0.5, 2.0, 64, 33), u(1e-4, 1.0, 64, 33))
big = u(20.0, 200.0, 50, 40)
rate = u(0.5, 4.0, 50, 40)
case("large_shape", big, rate, big / rate * u(0.7, 1.3, 50, 40))
case("one", torch.tensor([2.5]), torch.tensor([1.5]), torch.tensor([0.75]))
# the reference's 0 * log 0: shape exactly 1 at x = 0 gives NaN (torch.distributions' xlogy would give 0)
case("zero_x", torch.tensor([1.0, 2.0]), torch.tensor([1.0, 1.0]), torch.tensor([0.0, 1.0]))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gamma_vectors.npz"), **out)
print({k: (getattr(v, "shape", None), float(v) if np.ndim(v) == 0 else None) for k, v in out.items()})
