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


class BucketedGradSync:
  """Gradient all-reduce in two buckets, overlapped with the backward pass.

  The flat gradient buffer is laid out [encoder | generators, log_sigma, Ws]
  (engine.py).  The decoder kernel finishes the second part first, so its
  all-reduce (60 % of the bytes at the scale-up shape) is issued right after it and
  runs on the communicator's stream while the encoder backward GEMM computes; the
  encoder part follows.  `wait()` makes the compute stream wait for both and must
  be called before the optimizers and the proximal step;
  `wait_decoder()` / `wait_encoder()` wait for one bucket each, so that the decoder-side
  optimizers run under the encoder bucket's all-reduce (bench.py).
  Collectives are issued in the same order on every rank.  `holder` is anything
  with `flat_grads` and `seg_off` (an Engine)."""

  def __init__(self, holder, group=None, force=False):
    self.group = group
    self.active = force or is_active(group)
    flat = holder.flat_grads
    cut = holder.seg_off["gen_W1"]
    self.flat = flat
    self.enc = flat[:cut]
    self.dec = flat[cut:]
    self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    backend = dist.get_backend(group) if (dist.is_available() and dist.is_initialized()) else ""
    # RCCL averages in the collective; gloo has no AVG, so sum and scale once in wait()
    self._avg = backend == "nccl"
    self._pending = {"dec": [], "enc": []}
    self._unscaled = {"dec": False, "enc": False}

  def _issue(self, t, which):
    if not self.active or t.numel() == 0:
      return
    op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
    if not self._avg and t.is_cuda:
      # gloo with DEVICE tensors (ranks sharing a GPU in tests and rehearsals; never the production backend).  On this
      # stack (torch 2.10.0+rocm7.0) an async gloo all-reduce of a device tensor never completes -- or crawls, 19 s a
      # step -- when the tensor's stream still has work in flight at the moment the collective is ISSUED; with the
      # stream idle at that moment it takes its normal 20 ms.  Established with tools/gloo_hang_probe.py
      # (profiles/r03_gloo_hang_probe.txt): it needs no kernel of this library (one 1024^3 torch matmul in flight is
      # enough), no second process on the GPU (rank 1 on host tensors: same), nothing launched behind the issue, no
      # large buffer (4 KB: same) and no cold pinned staging; the same stream pattern without gloo (event, second --
      # also high-priority -- stream waits, copy to pinned memory, host-side synchronize) runs normally with one or
      # two processes.  So: ProcessGroupGloo's device path, not the engine's stream, not the decoder's LDS spin
      # waits.  Draining first is the one variant that works; the collective then runs synchronously.
      torch.cuda.synchronize(t.device)
      dist.all_reduce(t, op=op, group=self.group)
      self._unscaled[which] = True
      return
    self._pending[which].append(dist.all_reduce(t, op=op, group=self.group, async_op=True))
    if not self._avg:
      self._unscaled[which] = True

  def decoder_grads_ready(self):
    self._issue(self.dec, "dec")

  def encoder_grads_ready(self):
    self._issue(self.enc, "enc")

  def _wait(self, which):
    for w in self._pending[which]:
      w.wait()
    self._pending[which] = []
    if self._unscaled[which]:
      (self.dec if which == "dec" else self.enc).mul_(1.0 / self.world)
      self._unscaled[which] = False

  def wait_decoder(self):
    """The compute stream waits for the decoder-side bucket only: the optimizers of the generators, log_sigma and Ws
    (Engine.adam_step(part="dec"), sgd_prox_step) can then run while the encoder bucket is still being reduced."""
    self._wait("dec")

  def wait_encoder(self):
    self._wait("enc")

  def wait(self):
    self._wait("dec")
    self._wait("enc")
