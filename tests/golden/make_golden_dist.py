#!/usr/bin/env python3
"""Log-probabilities computed by the REAL reference's distribution classes
(/root/reference/lib/distributions.py) on seeded inputs:
    python tests/golden/make_golden_dist.py        (build container only)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from lib.distributions import Gamma, Normal, DistributionCat, _DistributionCat  # noqa: E402

g = torch.Generator().manual_seed(11)
out = {}
shape = torch.rand(7, 5, generator=g) * 4 + 0.2
rate = torch.rand(7, 5, generator=g) * 3 + 0.1
x = torch.rand(7, 5, generator=g) * 5 + 1e-3
out.update(gamma_shape=shape.numpy(), gamma_rate=rate.numpy(), gamma_x=x.numpy(),
           gamma_logprob=np.float32(Gamma(shape, rate).logprob(x).item()))
mu = torch.randn(7, 9, generator=g)
sig = torch.rand(7, 9, generator=g) + 0.05
xn = torch.randn(7, 9, generator=g)
out.update(normal_mu=mu.numpy(), normal_sigma=sig.numpy(), normal_x=xn.numpy(),
           normal_logprob=np.float32(Normal(mu, sig).logprob(xn).item()))
# mixed concatenation of Normals through the generic class (sizes 4 and 5 on the last dim)
cat = _DistributionCat([Normal(mu[:, :4], sig[:, :4]), Normal(mu[:, 4:], sig[:, 4:])])
out["cat_logprob"] = np.float32(cat.logprob(xn).item())
merged = DistributionCat([Normal(mu[:, :4], sig[:, :4]), Normal(mu[:, 4:], sig[:, 4:])])
out["cat_is_normal"] = np.int32(isinstance(merged, Normal))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_vectors.npz"), **out)
print({k: getattr(v, "shape", v) for k, v in out.items()})
