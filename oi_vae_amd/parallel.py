"""Data-parallel glue: one process per GPU, torch.distributed over RCCL.

The reference has no distributed code; the semantics are derived in SURVEY.md
section 8e.  Every loss term except the weight prior is a sum divided by the
local batch size (/root/reference/lib/oivae.py:50,55-59) and the weight prior is
batch independent, so with equal shards the exact global-batch gradient is the
MEAN over ranks of the local gradients.  One all-reduce over the engine's flat
gradient buffer, issued after backward and before the optimizers and the
proximal step, keeps parameters (and the zero set of Ws) identical on all ranks.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def is_active(group=None) -> bool:
  return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def allreduce_mean_(flat: torch.Tensor, group=None) -> torch.Tensor:
  """In-place mean over ranks of a flat buffer (sum all-reduce, then 1/R)."""
  if not is_active(group):
    return flat
  dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
  flat.mul_(1.0 / dist.get_world_size(group))
  return flat


def shard_rows(n_rows: int, rank: int, world: int):
  """Contiguous equal shards [lo, hi) of the batch; n_rows must divide evenly
  so that the mean over ranks is the exact global-batch gradient."""
  if n_rows % world:
    raise ValueError("global batch %d is not divisible by world size %d" % (n_rows, world))
  per = n_rows // world
  return rank * per, (rank + 1) * per


def reduce_scalars_(scalars: torch.Tensor, group=None) -> torch.Tensor:
  """Reported terms of one step, averaged over ranks.  z_kl and loglik_term are
  per-local-batch means, so their mean is the global value; logprob_theta and
  logprob_W are identical on all ranks and unchanged by the mean; loss and elbo
  are linear in those."""
  return allreduce_mean_(scalars, group)
