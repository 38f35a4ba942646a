#!/usr/bin/env python3
"""oi-VAE on MEG source estimates: counterpart of /root/reference/meg_oi_vae.py on the MI355X engine.

Groups are cortical labels with ragged vertex counts (`np.bincount(roiidx)`, meg_oi_vae.py:96-97), dim_z 10,
group_input_dim 1, Linear generators, batch 4096, lr 1e-3, Ws_lr 1e-6, lam 1 (meg_oi_vae.py:116-127,178,202,210).
The reference's data path needs `mne` and a private dataset (lib/meg_util.py, lib/sps_info.py); this driver takes
either --data FILE.npz (arrays `X` [frames, vertices] float32 already permuted so that each label's vertices are
contiguous, and `group_output_dims`) or synthetic frames with the same ragged structure.  With torchrun it runs
data-parallel: every rank takes its shard of each 4096-frame batch and gradients are averaged over RCCL before
the optimizers and the proximal step."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oi_vae_amd import parallel  # noqa: E402
from oi_vae_amd.experiments import Experiment, minibatches  # noqa: E402


def synthetic_meg(num_labels, frames, dim_z_true, seed):
  """Frames driven by a few latent sources, each reaching a subset of the labels; ragged label sizes."""
  rng = np.random.default_rng(seed)
  dims = rng.integers(40, 260, size=num_labels).tolist()
  D = int(sum(dims))
  Z = rng.standard_normal((frames, dim_z_true)).astype(np.float32)
  X = np.zeros((frames, D), np.float32)
  off = 0
  for d in dims:
    active = rng.random(dim_z_true) < 0.3
    A = rng.standard_normal((dim_z_true, d)).astype(np.float32) * active[:, None]
    X[:, off:off + d] = Z @ A + 0.1 * rng.standard_normal((frames, d)).astype(np.float32)
    off += d
  return X, dims


def main(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument("--data", default=None)
  ap.add_argument("--epochs", type=int, default=200)            # num_epochs, meg_oi_vae.py:125
  ap.add_argument("--max-iterations", type=int, default=0)
  ap.add_argument("--dim-z", type=int, default=10)
  ap.add_argument("--labels", type=int, default=44)
  ap.add_argument("--frames", type=int, default=32768)
  ap.add_argument("--log-every", type=int, default=50)
  ap.add_argument("--test-fraction", type=float, default=0.1,
                  help="frames held out for the test log-likelihood the reference leaves stubbed (meg_oi_vae.py:258-260)")
  ap.add_argument("--out", default="out_meg")
  args = ap.parse_args(argv)

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  torch.cuda.set_device(local_rank)
  if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

  torch.manual_seed(8675309)   # same seed on every rank: identical initial parameters and batch order
  if args.data:
    z = np.load(args.data)
    X, group_output_dims = z["X"].astype(np.float32), [int(d) for d in z["group_output_dims"]]
  else:
    X, group_output_dims = synthetic_meg(args.labels, args.frames, 6, seed=0)
  batch_size, mc_samples = 4096, 1
  exp = Experiment(dim_x=X.shape[1], dim_z=args.dim_z, group_dims=group_output_dims, group_input_dim=1,
                   gen_type="linear", lr=1e-3, Ws_lr=1e-6, lam=1.0, prior_rows=1,
                   device=torch.device("cuda", local_rank))
  n_test = int(X.shape[0] * args.test_fraction)
  Xtest = torch.from_numpy(X[X.shape[0] - n_test:]).to(exp.device) if n_test else None
  Xd = torch.from_numpy(X[:X.shape[0] - n_test]).to(exp.device)
  gen = torch.Generator().manual_seed(1)
  done = False
  for epoch in range(args.epochs):
    for Xbatch in minibatches(Xd, batch_size, generator=gen):
      if world > 1:  # equal shards; a ragged tail that does not divide is dropped from this step
        n = (Xbatch.size(0) // world) * world
        lo, hi = parallel.shard_rows(n, rank, world)
        Xbatch = Xbatch[lo:hi]
      info = exp.step(Xbatch, mc_samples)
      if rank == 0 and (exp.iteration - 1) % args.log_every == 0:
        exp.report(info, 'epoch %d iter %d' % (epoch, exp.iteration - 1))
      if args.max_iterations and exp.iteration >= args.max_iterations:
        done = True
        break
    if done:
      break
  # test log-likelihood: the held-out frames through the trained model, no gradient / optimizer / proximal side effect
  # (every rank evaluates the same frames in batches of 4096; per-frame average of log p(x | z), z ~ q(z | x))
  test_log_lik = None
  if Xtest is not None and Xtest.size(0) > 0:
    tot, cnt = 0.0, 0
    for lo in range(0, Xtest.size(0), batch_size):
      xb = Xtest[lo:lo + batch_size]
      tot += float(exp.vae.heldout_loglik(xb, mc_samples)) * xb.size(0)
      cnt += xb.size(0)
    test_log_lik = tot / cnt
  if rank == 0:
    C = exp.save_artifacts(args.out, 'meg')
    print('connectivity matrix %s, %d of %d connections exactly zero' % (C.shape, (C == 0).sum(), C.size))
    if test_log_lik is not None:
      print("test log-lik: %.4f per frame over %d held-out frames" % (test_log_lik, Xtest.size(0)))
      import json
      with open(os.path.join(args.out, 'meg_test_log_lik.json'), 'w') as f:
        json.dump({'test-log-lik': test_log_lik, 'frames': int(Xtest.size(0)), 'iterations': exp.iteration}, f)
  if world > 1:
    torch.distributed.destroy_process_group()
  return exp


if __name__ == "__main__":
  main()
