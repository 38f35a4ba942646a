"""Helpers shared by the tests: load a golden fixture (tests/golden/*.npz,
written by tests/golden/make_golden.py from the real reference) into the
oracle's parameter dict layout."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

STEP_CONFIGS = ["c1_bars_k8", "c1_bars_k16", "c1_bars_k16_strong", "c2_mocap", "c3_meg_like",
                "c4_tiny32", "c4_tiny64", "c4_mixed", "c5_wide", "c4_pad32", "c4_pad_m32", "c4_pad_dg32"]


def load(name):
  z = np.load(os.path.join(GOLDEN, name + ".npz"))
  spec = json.loads(str(z["spec"]))
  return z, spec


def params_at(z, spec, prefix, with_Ws=True):
  """prefix: 'init/' or 'stepN/post/' or 'stepN/grad/'; 'stepN/adam_m/' / 'stepN/adam_v/' (Adam's moments: no Ws)."""
  G = len(spec["group_dims"])
  p = {k: z[prefix + k] for k in ("enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig") + (("Ws",) if with_Ws else ())}
  p["gen_W2"] = [z[prefix + "gen_W2/%d" % g] for g in range(G)]
  p["gen_b2"] = [z[prefix + "gen_b2/%d" % g] for g in range(G)]
  p["log_sigma"] = [z[prefix + "log_sigma/%d" % g] for g in range(G)]
  if spec["gen_type"] == "mlp":
    p["gen_W1"] = [z[prefix + "gen_W1/%d" % g] for g in range(G)]
    p["gen_b1"] = [z[prefix + "gen_b1/%d" % g] for g in range(G)]
  return p


def flat_items(p):
  """Yield (name, array) over a params/grads dict, lists expanded."""
  for k, v in p.items():
    if isinstance(v, list):
      for g, a in enumerate(v):
        yield "%s/%d" % (k, g), np.asarray(a)
    else:
      yield k, np.asarray(v)


def rel_err(a, b):
  """max |a-b| / max(|b|_inf, tiny): error relative to the tensor's scale."""
  a = np.asarray(a, np.float64)
  b = np.asarray(b, np.float64)
  scale = max(float(np.max(np.abs(b))) if b.size else 0.0, 1e-30)
  return float(np.max(np.abs(a - b))) / scale if b.size else 0.0


def elementwise_within(a, b, rel=1e-5, floor=1e-3):
  """(fraction, count): of the elements of the reference tensor b whose magnitude is at least `floor` of the tensor's
  largest, the fraction at which a is within `rel` of THE ELEMENT ITSELF (not of the tensor's scale, as rel_err)."""
  a = np.asarray(a, np.float64).ravel()
  b = np.asarray(b, np.float64).ravel()
  if b.size == 0:
    return 1.0, 0
  big = np.abs(b) >= floor * np.max(np.abs(b))
  n = int(big.sum())
  if n == 0:
    return 1.0, 0
  ok = np.abs(a[big] - b[big]) <= rel * np.abs(b[big])
  return float(ok.mean()), n
