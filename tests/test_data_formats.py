"""Data formats either side of the hot path: the bars
sampler against images drawn by the reference under the same seeds, and the .amc
reader/writer against its grammar (the reference's lark parser cannot run here:
parity unpinned, see oi_vae_amd/mocap_data.py)."""
import os

import numpy as np
import pytest
import torch

from oi_vae_amd import bars_data, mocap_data

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SAMPLE = """#!OML:ASF F:\\VICON\\USERDATA\\INSTALL\\rory3\\rory3.ASF
:FULLY-SPECIFIED
:DEGREES
1
root 9.37216 17.8693 -17.3198 -2.01677 -7.59696 -3.23164
lowerback 2.30193 -0.395121 1.17299
rhumerus -29.5133 -12.8185 -82.7731
rradius 1.2e+1
# a comment line
2
root 9.37285 17.8666 -17.3192 -2.06376 -7.58832 -3.1009
lowerback 2.23572 -0.406032 1.14828
rhumerus -29.5133 -12.8185 -82.7731
rradius -.5
"""


def test_bars_sampler_draws_the_reference_images():
  z = np.load(os.path.join(GOLDEN, "bars_samples.npz"))
  torch.manual_seed(5)
  one = torch.stack([bars_data.sample_one_bar_image(8).view(-1) for _ in range(16)]).numpy()
  assert np.array_equal(one, z["one_bar_8"])
  assert set(np.unique(one)) == {0.0, 0.5} and np.all((one.reshape(16, 8, 8) != 0).any(2).sum(1) == 1)
  torch.manual_seed(6)
  assert np.array_equal(bars_data.sample_many_bars_images(12, 16, 2.0, 0.0).numpy(), z["many_16"])
  torch.manual_seed(7)
  assert np.array_equal(bars_data.sample_bars_image(torch.full((5,), 0.4), torch.full((7,), 0.3)).numpy(), z["image_5x7"])


def test_amc_parse_sample():
  frames = mocap_data.parse_amc_data(SAMPLE)
  assert [f[0] for f in frames] == [1, 2]
  assert list(frames[0][1].keys()) == ["root", "lowerback", "rhumerus", "rradius"]
  assert frames[0][1]["root"] == [9.37216, 17.8693, -17.3198, -2.01677, -7.59696, -3.23164]
  assert frames[0][1]["rradius"] == [12.0] and frames[1][1]["rradius"] == [-0.5]
  dims, arr = mocap_data.amc_to_array(frames, ["root", "rradius", "lowerback", "rhumerus"])
  assert dims == {"root": 6, "rradius": 1, "lowerback": 3, "rhumerus": 3}
  assert len(arr) == 2 and len(arr[0]) == 13 and arr[1][6] == -0.5


@pytest.mark.parametrize("bad", [
  "1\nroot 1 2 3\n",                                        # no preamble
  ":FULLY-SPECIFIED\n:DEGREES\nroot 1 2\n",                  # measurement before a key frame
  ":FULLY-SPECIFIED\n:DEGREES\n1\nroot 1 x 3\n",             # not a number
  ":FULLY-SPECIFIED\n:DEGREES\n1\nroot\n",                   # joint without channels
  ":FULLY-SPECIFIED\n:DEGREES\n1\n2\nroot 1\n",              # empty key frame
])
def test_amc_rejects_text_outside_the_grammar(bad):
  with pytest.raises(ValueError):
    mocap_data.parse_amc_data(bad)


def test_amc_array_checks_frame_numbering_and_joint_sets():
  frames = mocap_data.parse_amc_data(SAMPLE)
  with pytest.raises(AssertionError):
    mocap_data.amc_to_array(frames[1:], ["root", "lowerback", "rhumerus", "rradius"])   # ids must start at 1
  with pytest.raises(AssertionError):
    mocap_data.amc_to_array(frames, ["root", "lowerback", "rhumerus"])                  # joint sets must agree


def test_amc_write_read_round_trip(tmp_path):
  rng = np.random.default_rng(0)
  joint_order = ["root", "lowerback", "lhand", "rtoes"]
  joint_dims = {"root": 3, "lowerback": 3, "lhand": 2, "rtoes": 1}   # root: rotations only, as in the script
  seq = rng.standard_normal((7, 9)).astype(np.float32) * 50
  d = tmp_path / "cmu_mocap" / "subjects" / "07"
  d.mkdir(parents=True)
  path = mocap_data.subject_trial_path(7, 2, datadir=str(tmp_path))
  assert path.endswith(os.path.join("cmu_mocap", "subjects", "07", "07_02.amc"))
  mocap_data.save_amc(path, torch.from_numpy(seq), joint_order, joint_dims)
  order, dims, arr = mocap_data.load_mocap_trial(7, 2, datadir=str(tmp_path))
  assert order == joint_order and dims == {"root": 6, "lowerback": 3, "lhand": 2, "rtoes": 1}
  arr = np.asarray(arr, np.float32)
  assert arr.shape == (7, 12) and np.all(arr[:, :3] == 0)            # translations written as zeros
  assert np.array_equal(arr[:, 3:], seq)                              # %.9g round-trips fp32 exactly


def test_bars_probabilities_keep_the_reference_association():
  """n = 7, mul = 3: fl(fl(1/7) * 3) != fl(3/7).  The Bernoulli draws compare uniform numbers against the probability,
  so the stream only equals the reference's (lib/bars_data.py:16-23) with its association of the two roundings."""
  import torch
  p_ref = torch.ones(7) / 7 * 3.0
  assert float(p_ref[0]) != float(torch.tensor(3.0 / 7))
  torch.manual_seed(5)
  want = torch.stack([0.5 * (torch.bernoulli(p_ref).unsqueeze(1) + torch.bernoulli(p_ref).unsqueeze(0)).reshape(-1)
                      for _ in range(200)])
  torch.manual_seed(5)
  assert torch.equal(bars_data.sample_many_bars_images(200, 7, 3.0, 3.0), want)


def test_kl_normals_keeps_the_two_logarithms_apart():
  import torch
  from oi_vae_amd.distributions import Normal
  from oi_vae_amd.utils import KL_Normals
  d0 = Normal(torch.zeros(1, 2), torch.tensor([[1e-30, 1e25]]))
  d1 = Normal(torch.zeros(1, 2), torch.tensor([[1e25, 1e-30]]))
  assert torch.isinf(torch.log(d1.sigma / d0.sigma)).any()      # the quotient leaves fp32's range ...
  assert torch.isfinite(torch.log(d1.sigma) - torch.log(d0.sigma)).all()
  mu, s = torch.randn(3, 4), torch.rand(3, 4) + 0.1
  got = KL_Normals(Normal(mu, s), Normal(torch.zeros(3, 4), torch.ones(3, 4)))
  want = torch.sum(-0.5 + (torch.pow(s, 2) + torch.pow(mu, 2)) / 2 + 0.0 - torch.log(s))
  assert got.item() == want.item()


def test_distribution_classes_match_reference_vectors():
  """Gamma / Normal / _DistributionCat log-probabilities against values recorded from the
  reference's classes (tests/golden/make_golden_dist.py) and, for Gamma, scipy's closed form."""
  import os
  import scipy.stats
  import torch
  from oi_vae_amd.distributions import DistributionCat, Gamma, Normal, _DistributionCat
  v = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dist_vectors.npz"))
  t = lambda k: torch.from_numpy(v[k])
  got = Gamma(t("gamma_shape"), t("gamma_rate")).logprob(t("gamma_x")).item()
  # same four fp32 terms in the same order as the reference: the same bits; float64 scipy below
  assert got == float(v["gamma_logprob"])
  want = scipy.stats.gamma.logpdf(v["gamma_x"].astype(np.float64), v["gamma_shape"].astype(np.float64),
                                  scale=1.0 / v["gamma_rate"].astype(np.float64)).sum()
  assert abs(got - want) < 1e-4 * abs(want)
  mu, sig, x = t("normal_mu"), t("normal_sigma"), t("normal_x")
  assert Normal(mu, sig).logprob(x).item() == float(v["normal_logprob"])
  parts = [Normal(mu[:, :4], sig[:, :4]), Normal(mu[:, 4:], sig[:, 4:])]
  assert _DistributionCat(parts).logprob(x).item() == float(v["cat_logprob"])
  assert isinstance(DistributionCat(parts), Normal) == bool(v["cat_is_normal"])
  mixed = DistributionCat([parts[0], Gamma(t("gamma_shape"), t("gamma_rate"))])
  assert isinstance(mixed, _DistributionCat)
  xm = torch.cat([x[:, :4], t("gamma_x")], dim=1)
  assert abs(mixed.logprob(xm).item() - (parts[0].logprob(x[:, :4]).item() + got)) < 1e-3


def test_gamma_host_expression_matches_the_reference_in_every_regime():
  """Gamma.logprob on host tensors (the torch expression in the reference's term order) against the values recorded
  from the reference's class in four regimes (tests/golden/make_golden_gamma.py); on this torch build: the same bits."""
  import torch
  from oi_vae_amd.distributions import Gamma
  z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gamma_vectors.npz"))
  for name in ("typical", "small_shape", "large_shape", "one"):
    a, b, x = (torch.tensor(z[name + "/" + k], requires_grad=True) for k in ("shape", "rate", "x"))
    lp = Gamma(a, b).logprob(x)
    ref = float(z[name + "/logprob"])
    assert abs(lp.item() - ref) <= 1e-6 * abs(ref), name
    lp.backward()
    for k, t in (("d_shape", a), ("d_rate", b), ("d_x", x)):
      r = z[name + "/" + k]
      assert np.abs(t.grad.numpy() - r).max() <= 1e-5 * np.abs(r).max(), (name, k)
  a, b, x = (torch.tensor(z["zero_x/" + k]) for k in ("shape", "rate", "x"))
  assert torch.isnan(Gamma(a, b).logprob(x)).item() and np.isnan(float(z["zero_x/logprob"]))
