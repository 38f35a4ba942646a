"""Pins tests/_f64_reference.py -- the float64 torch statement of the step that the full-size GPU tests compare the
HIP path with -- against the golden vectors recorded from the real reference: every step config, first step,
scalars and every gradient.  CPU only (the same code runs on the GPU at sizes the CPU cannot reach)."""
import numpy as np
import pytest
import torch

from oi_vae_amd.engine import ModelSpec
from tests._f64_reference import f64_step
from tests._fixtures import STEP_CONFIGS, flat_items, load, params_at, rel_err

SCALARS = ("z_kl", "loglik_term", "logprob_theta", "logprob_W", "loss", "elbo")


class HostHolder:
  """What f64_step reads of an Engine (spec, p, group_off_host), on the CPU."""

  def __init__(self, spec, params):
    self.spec = ModelSpec(D=spec["D"], K=spec["K"], H=spec["H"], group_dims=spec["group_dims"],
                          gen_type=spec["gen_type"], M=spec.get("M", 0))
    self.group_off_host = np.concatenate([[0], np.cumsum(spec["group_dims"])]).astype(np.int32)
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    self.p = {k: t(params[k]) for k in ("enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig", "Ws")}
    for k in ("gen_W2", "gen_b2", "log_sigma"):
      self.p[k] = t(np.concatenate([np.asarray(a, np.float32) for a in params[k]], axis=0))
    if spec["gen_type"] == "mlp":
      self.p["gen_W1"] = t(np.stack(params["gen_W1"]))
      self.p["gen_b1"] = t(np.stack(params["gen_b1"]))


def split_like(spec, grads):
  offs = np.concatenate([[0], np.cumsum(spec["group_dims"])])
  G = len(spec["group_dims"])
  out = {k: grads[k].numpy() for k in ("enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig", "Ws")}
  for k in ("gen_W2", "gen_b2", "log_sigma"):
    out[k] = [grads[k][offs[g]:offs[g + 1]].numpy() for g in range(G)]
  if spec["gen_type"] == "mlp":
    out["gen_W1"] = [grads["gen_W1"][g].numpy() for g in range(G)]
    out["gen_b1"] = [grads["gen_b1"][g].numpy() for g in range(G)]
  return out


@pytest.mark.parametrize("name", STEP_CONFIGS + ["c4_trained64"])
def test_f64_reference_matches_reference_golden(name):
  z, spec = load(name)
  steps = [0] if name != "c4_trained64" else [0]
  for i in steps:
    pre = "step%d/" % i
    holder = HostHolder(spec, params_at(z, spec, "init/"))
    X, eps = torch.from_numpy(z[pre + "X"]), torch.from_numpy(z[pre + "eps"])
    sc, grads = f64_step(holder, X, eps, float(z[pre + "c"]), sigma_theta=spec["sigma_theta"], lam=spec["lam"],
                         rows_per_chunk=23)   # ragged chunks: the chunked accumulation is what the big sizes use
    for k in SCALARS:
      want = float(z[pre + k])
      assert abs(sc[k] - want) <= 2e-6 * max(abs(want), 1.0), (name, k, sc[k], want)
    got = dict(flat_items(split_like(spec, grads)))
    for k, b in flat_items(params_at(z, spec, pre + "grad/")):
      assert rel_err(got[k], b) < 1e-5, (name, k, rel_err(got[k], b))


@pytest.mark.parametrize("i", [12, 29])
def test_f64_reference_on_the_trained_state(i):
  """... and from the reference's own state late in the sparse run (dead rows, moved log_sigma, S = 2 at step 29)."""
  z, spec = load("c4_trained64")
  prev, pre = "step%d/" % (i - 1), "step%d/" % i
  holder = HostHolder(spec, params_at(z, spec, prev + "post/"))
  sc, grads = f64_step(holder, torch.from_numpy(z[pre + "X"]), torch.from_numpy(z[pre + "eps"]), float(z[pre + "c"]),
                       sigma_theta=spec["sigma_theta"], lam=spec["lam"], rows_per_chunk=16)
  for k in SCALARS:
    want = float(z[pre + k])
    assert abs(sc[k] - want) <= 2e-6 * max(abs(want), 1.0), (k, sc[k], want)
  got = dict(flat_items(split_like(spec, grads)))
  for k, b in flat_items(params_at(z, spec, pre + "grad/")):
    assert rel_err(got[k], b) < 1e-5, (k, rel_err(got[k], b))
