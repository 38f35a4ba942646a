#!/usr/bin/env python3
"""oi-VAE on the bars data: counterpart of /root/reference/bars_data_oivae.py on the MI355X engine.

8x8 one-bar images, one group per image row, group_input_dim 1, Linear generators, prior_z of batch shape,
lr 1e-2, Ws_lr 1e-4, lam 1, random minibatches of 64 out of 2048 (bars_data_oivae.py:20-44,212-268).
Writes the connectivity matrix instead of showing plots."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oi_vae_amd.bars_data import sample_one_bar_image  # noqa: E402
from oi_vae_amd.experiments import Experiment  # noqa: E402


def main(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument("--iterations", type=int, default=20000)      # num_epochs, bars_data_oivae.py:30
  ap.add_argument("--dim-z", type=int, default=8)               # image_size in the script; BASELINE.json runs 16 too
  ap.add_argument("--lam", type=float, default=1.0)
  ap.add_argument("--log-every", type=int, default=500)
  ap.add_argument("--torch-optimizers", action="store_true", help="caller-owned torch.optim instead of the fused kernels")
  ap.add_argument("--out", default="out_bars")
  args = ap.parse_args(argv)

  torch.manual_seed(0)
  image_size, num_train_samples, batch_size, mc_samples = 8, 2048, 64, 1
  X = torch.stack([sample_one_bar_image(image_size).view(-1) for _ in range(num_train_samples)])
  X += 0.05 * torch.randn(X.size())
  exp = Experiment(dim_x=image_size * image_size, dim_z=args.dim_z, group_dims=[image_size] * image_size,
                   group_input_dim=1, gen_type="linear", lr=1e-2, Ws_lr=1e-4, lam=args.lam,
                   prior_rows=batch_size, native_optimizers=not args.torch_optimizers)
  Xd = X.to(exp.device)  # resident on the GPU: no per-step host copy
  for i in range(args.iterations):
    idx = torch.randperm(num_train_samples)[:batch_size].to(exp.device)
    info = exp.step(Xd[idx], mc_samples)
    if i % args.log_every == 0 or i == args.iterations - 1:
      exp.report(info, 'iter %d' % i)
  C = exp.save_artifacts(args.out, 'bars_data')
  print('connectivity matrix %s, %d of %d connections exactly zero -> %s' % (C.shape, (C == 0).sum(), C.size, args.out))
  return exp


if __name__ == "__main__":
  main()
