#!/usr/bin/env python3
"""Golden vectors for FirstLayerSparseDecoder.proximal_step, recorded from the REAL
reference (/root/reference/lib/models.py:29-71) in the build container:

    python tests/golden/make_golden_fls.py

Each case builds the reference's decoder with `torch.nn.Linear(1, 1, bias=False)`
stand-in generators (proximal_step never touches them), overwrites the
latent-to-group maps with recorded weights, calls the unmodified
`proximal_step(t)` and stores inputs and outputs.  Only numbers are written.
Cases cover uniform and ragged group input dims, K (input_dim) on every side of
the 4/8/32-column boundaries of torch's column-sum kernel, thresholds at exact
column norms and their neighbours, zero and denormal columns.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from lib.models import FirstLayerSparseDecoder  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def run_case(dims, K, t, W_list):
  dec = FirstLayerSparseDecoder([torch.nn.Linear(1, 1, bias=False) for _ in dims], list(dims), K)
  for m, W in zip(dec.latent_to_group_maps, W_list):
    assert tuple(m.weight.shape) == W.shape
    m.weight.data.copy_(torch.from_numpy(W))
  pen = float(dec.group_lasso_penalty().item())
  dec.proximal_step(float(t))
  return [m.weight.data.numpy().copy() for m in dec.latent_to_group_maps], pen


def main():
  torch.manual_seed(0)
  torch.set_num_threads(1)
  rng = np.random.default_rng(0)
  out = {}
  cases = []
  small = ([1] * 6, [1, 2, 3, 4, 5, 8], [9, 15, 16, 17, 20])
  big = ([31, 32, 33, 48], [63, 64, 65, 100], [128, 255])
  for K in (1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 32, 33, 40, 65):
    for dims in small:
      cases.append((K, dims))
  for K in (1, 3, 8, 33):
    for dims in big:
      cases.append((K, dims))
  names = []
  for ci, (K, dims) in enumerate(cases):
    W = [(rng.standard_normal((h, K)) * rng.uniform(0.05, 3.0)).astype(np.float32) for h in dims]
    # adversarial columns: exact zero, denormal, huge
    if K >= 3:
      W[0][:, 0] = 0.0
      W[-1][:, 1] = np.float32(1e-41)
      W[-1][:, 2] *= np.float32(1e15)
    norms = np.concatenate([np.sqrt((w.astype(np.float64) ** 2).sum(0)) for w in W])
    t_med = np.float32(np.median(norms))
    # thresholds: median, an exact fp32 column norm as torch computes it and its upper neighbour, tiny
    exact = torch.sqrt(torch.sum(torch.pow(torch.from_numpy(W[len(W) // 2]), 2), dim=0)).numpy()[K // 2]
    for ti, t in enumerate((t_med, exact, np.nextafter(exact, np.float32(np.inf)), np.float32(1e-6))):
      res, pen = run_case(dims, K, t, W)
      name = "c%d_t%d" % (ci, ti)
      names.append(name)
      out[name + "/K"] = np.int32(K)
      out[name + "/dims"] = np.asarray(dims, np.int32)
      out[name + "/t"] = np.float32(t)
      out[name + "/pen"] = np.float32(pen)
      if ti == 0:
        out["c%d/W" % ci] = np.concatenate(W, axis=0)
      out[name + "/out"] = np.concatenate(res, axis=0)
  out["names"] = np.asarray(names)
  np.savez_compressed(os.path.join(HERE, "fls_prox_vectors.npz"), **out)
  print("wrote", len(names), "cases")


def main_tall():
  """Maps of 272 rows and more (fls_prox_vectors_tall.npz): from 272 rows on the column sum's cascade uses its third
  level (a flush every 256 rows), from 4112 on its fourth; H_g * K stays under 32768, where torch reduces in one
  thread."""
  torch.manual_seed(0)
  torch.set_num_threads(1)
  rng = np.random.default_rng(5)
  out = {}
  names = []
  cases = [(3, [272, 300]), (8, [511, 257]), (32, [600, 288]), (33, [300, 272, 400]), (40, [700]), (31, [1000]),
           (7, [4097, 4112]), (2, [9000]), (5, [4200])]
  for ci, (K, dims) in enumerate(cases):
    W = [(rng.standard_normal((h, K)) * rng.uniform(0.05, 3.0)).astype(np.float32) for h in dims]
    if K >= 3:
      W[0][:, 0] = 0.0
      W[-1][:, 1] = np.float32(1e-41)
    norms = np.concatenate([np.sqrt((w.astype(np.float64) ** 2).sum(0)) for w in W])
    t_med = np.float32(np.median(norms))
    exact = torch.sqrt(torch.sum(torch.pow(torch.from_numpy(W[len(W) // 2]), 2), dim=0)).numpy()[K // 2]
    for ti, t in enumerate((t_med, exact, np.nextafter(exact, np.float32(np.inf)))):
      res, pen = run_case(dims, K, t, W)
      name = "c%d_t%d" % (ci, ti)
      names.append(name)
      out[name + "/K"] = np.int32(K)
      out[name + "/dims"] = np.asarray(dims, np.int32)
      out[name + "/t"] = np.float32(t)
      out[name + "/pen"] = np.float32(pen)
      if ti == 0:
        out["c%d/W" % ci] = np.concatenate(W, axis=0)
      out[name + "/out"] = np.concatenate(res, axis=0)
  out["names"] = np.asarray(names)
  np.savez_compressed(os.path.join(HERE, "fls_prox_vectors_tall.npz"), **out)
  print("wrote", len(names), "tall cases")


if __name__ == "__main__":
  if "tall" in sys.argv[1:]:
    main_tall()
  else:
    main()
