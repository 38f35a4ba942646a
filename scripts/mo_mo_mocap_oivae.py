#!/usr/bin/env python3
"""oi-VAE on CMU mocap poses: counterpart of /root/reference/mo_mo_mocap_oivae.py on the MI355X engine.

29 one-joint groups, 59 channels (root translation dropped), dim_z 8, group_input_dim 8, Tanh -> Linear
generators, mc_samples 10, batch 64 with a ragged last batch, per-channel min-max normalisation
(mo_mo_mocap_oivae.py:26-35,68-98,133-155,195-212).  Reads ./data/cmu_mocap/subjects/07/07_NN.amc when present
(the dataset is not redistributed); otherwise --synthetic poses of the same layout drive the same code.
Writes the per-latent normalised connectivity matrix and the reconstructed sequence as an .amc file."""
import argparse
import itertools
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oi_vae_amd import mocap_data  # noqa: E402
from oi_vae_amd.experiments import Experiment, minibatches  # noqa: E402

GROUPS = [[j] for j in (
  'root', 'lowerback', 'upperback', 'thorax', 'lowerneck', 'upperneck', 'head', 'rclavicle', 'rhumerus', 'rradius',
  'rwrist', 'rhand', 'rfingers', 'rthumb', 'lclavicle', 'lhumerus', 'lradius', 'lwrist', 'lhand', 'lfingers', 'lthumb',
  'rfemur', 'rtibia', 'rfoot', 'rtoes', 'lfemur', 'ltibia', 'lfoot', 'ltoes')]
# channels per joint in the CMU skeleton; root has 3 translations + 3 rotations
CMU_DIMS = dict(root=6, lowerback=3, upperback=3, thorax=3, lowerneck=3, upperneck=3, head=3, rclavicle=2, rhumerus=3,
                rradius=1, rwrist=1, rhand=2, rfingers=1, rthumb=2, lclavicle=2, lhumerus=3, lradius=1, lwrist=1,
                lhand=2, lfingers=1, lthumb=2, rfemur=3, rtibia=1, rfoot=2, rtoes=1, lfemur=3, ltibia=1, lfoot=2, ltoes=1)


def synthetic_trials(joint_order, n_trials, frames, seed):
  """Smooth quasi-periodic joint angles with the CMU channel layout (a stand-in for the walking trials)."""
  g = torch.Generator().manual_seed(seed)
  D = sum(CMU_DIMS[j] for j in joint_order)
  out = []
  for _ in range(n_trials):
    t = torch.arange(frames, dtype=torch.float32).unsqueeze(1) / 120.0
    freq = 0.5 + torch.rand(1, D, generator=g)
    phase = 2 * math.pi * torch.rand(1, D, generator=g)
    amp = 30 * torch.rand(1, D, generator=g)
    arr = amp * torch.sin(2 * math.pi * freq * t + phase) + torch.randn(frames, D, generator=g)
    out.append((joint_order, dict(CMU_DIMS), arr.tolist()))
  return out


def main(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument("--epochs", type=int, default=50)
  ap.add_argument("--max-iterations", type=int, default=0)
  ap.add_argument("--dim-z", type=int, default=8)
  ap.add_argument("--synthetic", action="store_true")
  ap.add_argument("--log-every", type=int, default=200)
  ap.add_argument("--out", default="out_mocap")
  args = ap.parse_args(argv)

  torch.manual_seed(0)
  lam, mc_samples, batch_size, group_input_dim = 1.0, 10, 64, 8
  joint_order = [joint for grp in GROUPS for joint in grp]
  train_trials = [(7, t) for t in range(1, 11)]
  have_data = os.path.exists(mocap_data.subject_trial_path(*train_trials[0]))
  if args.synthetic or not have_data:
    if not args.synthetic:
      print('no CMU data under %s: using synthetic poses' % mocap_data.DATADIR)
    trials = synthetic_trials(joint_order, 4, 300, seed=1)
  else:
    trials = [mocap_data.load_mocap_trial(s, t, joint_order=joint_order) for s, t in train_trials]
  _, joint_dims, _ = trials[0]
  joint_dims = dict(joint_dims)
  joint_dims['root'] = joint_dims['root'] - 3    # root position in space is not learned (mo_mo_mocap_oivae.py:131-137)
  Xtrain_raw = torch.FloatTensor(list(itertools.chain(*[arr for _, _, arr in trials])))[:, 3:]
  mins, _ = torch.min(Xtrain_raw, dim=0)
  maxs, _ = torch.max(Xtrain_raw, dim=0)
  Xtrain = (Xtrain_raw - mins) / torch.clamp(maxs - mins, min=0.1)
  group_dims = [sum(joint_dims[j] for j in grp) for grp in GROUPS]

  exp = Experiment(dim_x=Xtrain.size(1), dim_z=args.dim_z, group_dims=group_dims, group_input_dim=group_input_dim,
                   gen_type="tanh_linear", lr=1e-3, Ws_lr=1e-4, lam=lam, prior_rows=1)
  Xd = Xtrain.to(exp.device)
  done = False
  for epoch in range(args.epochs):
    for Xbatch in minibatches(Xd, batch_size):
      info = exp.step(Xbatch, mc_samples)
      if (exp.iteration - 1) % args.log_every == 0:
        exp.report(info, 'epoch %d iteration %d' % (epoch, exp.iteration - 1))
      if args.max_iterations and exp.iteration >= args.max_iterations:
        done = True
        break
    if done:
      break
  names = [g[0] for g in GROUPS]
  C = exp.save_artifacts(args.out, 'mocap', group_names=names, normalise_per_latent=True)
  print('connectivity matrix %s, %d of %d connections exactly zero' % (C.shape, (C == 0).sum(), C.size))
  # reconstruction of the training sequence back in degrees (mo_mo_mocap_oivae.py:353-354)
  with torch.no_grad():
    recon = exp.generative_net(exp.inference_net(Xd).mu).mu.cpu() * torch.clamp(maxs - mins, min=0.1) + mins
  mocap_data.save_amc(os.path.join(args.out, '07_reconstructed.amc'), recon, joint_order, joint_dims)
  return exp


if __name__ == "__main__":
  main()
