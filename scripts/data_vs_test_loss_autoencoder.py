#!/usr/bin/env python3
"""Test loss against training-set size for the group-lasso AUTOENCODER variant: counterpart of
/root/reference/data_vs_test_loss_autoencoder.py on the MI355X.

Tied-bars images (16 x 16, the left half repeated on the right), a bias-free linear encoder to dim_z = 8 and a
FirstLayerSparseDecoder with one group per image row (group input 1; Linear -> Tanh -> Linear -> Tanh -> Linear
generators), Adam at 1e-2 on everything, then `decoder.proximal_step(lr * lam)` -- the one operation of this variant
that has a hand-written kernel (oivae_prox_cols_f32, bit-identical to the reference's CPU arithmetic); forward and
backward are ordinary torch modules on the GPU, as the reference trains them with autograd
(data_vs_test_loss_autoencoder.py:16-50,82-145).  For every (training-set size, lambda) pair the per-epoch test loss,
train loss and sparsity penalty are kept; the result goes to a JSON file (the reference pickles the same structure)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oi_vae_amd.bars_data import sample_bars_image  # noqa: E402
from oi_vae_amd.models import FirstLayerSparseDecoder  # noqa: E402


def tied_bars(n, image_size, repeats):
  """n images whose `repeats` vertical stripes are copies of one bars image of width image_size / repeats: rows drawn
  with probability 2 / image_size each, no columns (data_vs_test_loss_autoencoder.py:66-74)."""
  rows = torch.ones(image_size // repeats) / image_size * 2
  cols = torch.ones(image_size) / image_size * 0
  part = torch.stack([sample_bars_image(rows, cols).view(-1) for _ in range(n)])
  return torch.cat([part] * repeats, dim=1)


def nonlinear_decoder(image_size, dim_z):
  def generator():
    return torch.nn.Sequential(torch.nn.Linear(1, image_size, bias=False), torch.nn.Tanh(),
                               torch.nn.Linear(image_size, image_size, bias=False), torch.nn.Tanh(),
                               torch.nn.Linear(image_size, image_size, bias=False))
  return FirstLayerSparseDecoder([generator() for _ in range(image_size)], [1] * image_size, dim_z)


def linear_decoder(image_size, dim_z):
  return FirstLayerSparseDecoder([torch.nn.Linear(1, image_size, bias=False) for _ in range(image_size)],
                                 [1] * image_size, dim_z)


def reconstruction_loss(encoder, decoder, X):
  out = decoder(encoder(X))
  return torch.sum(torch.pow(out - X, 2)) / X.size(0), out


def main(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument("--epochs", type=int, default=2500)                       # num_epochs, :21
  ap.add_argument("--train-sizes", type=int, nargs="+", default=[2, 4, 8, 16, 32, 64, 128, 256, 512])
  ap.add_argument("--lams", type=float, nargs="+", default=[0.0, 0.1])     # :24
  ap.add_argument("--test-samples", type=int, default=2048)
  ap.add_argument("--linear-decoder", action="store_true")
  ap.add_argument("--out", default="out_autoencoder/data_vs_test_loss.json")
  args = ap.parse_args(argv)
  if not torch.cuda.is_available():
    raise SystemExit("this script trains on the GPU (the proximal step has no CPU path)")
  dev = torch.device("cuda", 0)
  torch.manual_seed(0)
  image_size, lr = 16, 1e-2
  dim_z = image_size // 2
  Xtest = tied_bars(args.test_samples, image_size, 2).to(dev)
  results = []
  for n_train in args.train_sizes:
    Xtrain = tied_bars(n_train, image_size, 2).to(dev)
    for lam in args.lams:
      encoder = torch.nn.Linear(image_size * image_size, dim_z, bias=False).to(dev)
      decoder = (linear_decoder if args.linear_decoder else nonlinear_decoder)(image_size, dim_z).cuda()
      opt = torch.optim.Adam([{"params": encoder.parameters(), "lr": lr}, {"params": list(decoder.parameters()), "lr": lr}])
      curves = torch.zeros(args.epochs, 3, device=dev)   # per epoch: test loss, train loss, lam * penalty (kept on the device)
      collapsed = 0
      for ep in range(args.epochs):
        with torch.no_grad():
          test_loss, rec = reconstruction_loss(encoder, decoder, Xtest)
          collapsed += int(ep % 500 == 0 and float(torch.sum(torch.abs(rec - rec[0]))) / Xtest.size(0) <= 1e-3)
          curves[ep, 0] = test_loss
          curves[ep, 2] = lam * decoder.group_lasso_penalty()
        train_loss, _ = reconstruction_loss(encoder, decoder, Xtrain)
        curves[ep, 1] = train_loss.detach()
        opt.zero_grad()
        train_loss.backward()
        opt.step()
        decoder.proximal_step(lr * lam)
      c = curves.cpu()
      zero = int((decoder.connectivity() == 0).sum())
      print("train samples %4d  lam %-5g  final test loss %.4f  train loss %.4f  zero connections %d of %d%s"
            % (n_train, lam, float(c[-1, 0]), float(c[-1, 1]), zero, image_size * dim_z,
               "  (solution collapsed at a check)" if collapsed else ""), flush=True)
      results.append({"num_train_samples": n_train, "lam": lam, "test_loss": c[:, 0].tolist(), "train_loss": c[:, 1].tolist(),
                      "sparsity_penalty": c[:, 2].tolist(), "zero_connections": zero})
  os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
  with open(args.out, "w") as fh:
    json.dump({"results": results, "nums_train_samples": args.train_sizes, "lams": args.lams, "epochs": args.epochs}, fh)
  print("wrote", args.out)
  return results


if __name__ == "__main__":
  main()
