"""GPU tests at sizes the oracle cannot reach, through size-independent
properties: (1) the MFMA kernels agree with the shape-generic kernels on the
same inputs; (2) at the BASELINE scale-up shape (G=1024 x 64, K=64, batch 65536)
the full-batch step equals the mean of its two half-batch steps (the identity
the data-parallel all-reduce relies on); (3) the proximal step at the wide-prox
shape (G=8192, K=256) is idempotent on its zero set and bit-equal to the
oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oi_vae_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(G, dg, K, H, M, seed=0):
  from oi_vae_amd import Engine, ModelSpec
  spec = ModelSpec(D=G * dg, K=K, H=H, M=M, group_dims=[dg] * G, gen_type="mlp")
  eng = Engine(spec, device="cuda:0")
  gen = torch.Generator(device="cuda").manual_seed(seed)
  for n in ("enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig"):
    eng.p[n].copy_((torch.rand(eng.p[n].shape, generator=gen, device="cuda") * 2 - 1) / (G * dg) ** 0.5)
  for n in ("gen_W1", "gen_b1", "gen_W2", "gen_b2"):
    eng.p[n].copy_((torch.rand(eng.p[n].shape, generator=gen, device="cuda") * 2 - 1) / 8)
  eng.p["log_sigma"].fill_(float(np.log(1e-2)))
  eng.p["Ws"].copy_(torch.randn(eng.p["Ws"].shape, generator=gen, device="cuda"))
  return eng


_CROSS = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from tests.test_gpu_scale import _engine
mode = sys.argv[1]
eng = _engine(24, 64, 64, 64, 64, seed=1)
g = torch.Generator(device="cuda").manual_seed(5)
B, S = 200, 2   # ragged: 400 rows = 12.5 wave tiles
X = torch.rand(B, 24 * 64, generator=g, device="cuda")
eps = torch.randn(S, B, 64, generator=g, device="cuda")
out = eng.forward_backward(X, eps, 0.7)
torch.cuda.synchronize()
np.savez(sys.argv[2], grads=eng.flat_grads.cpu().numpy(), sc=out["scalars"].cpu().numpy(), z=out["z"].cpu().numpy())
'''


def test_mfma_kernels_agree_with_generic_kernels(tmp_path):
  script = tmp_path / "cross.py"
  script.write_text(_CROSS % {"root": ROOT})
  res = {}
  for mode in ("mfma", "generic", "layerwise", "fp32", "onewave", "pair", "bf16"):
    env = dict(os.environ)
    if mode == "generic":
      env["OIVAE_FORCE_GENERIC"] = "1"
    if mode == "layerwise":
      env["OIVAE_DEC"] = "v1"   # the nine-product fp32 template kernel
    if mode == "fp32":
      env["OIVAE_DEC"] = "v4"   # the fp32-MFMA composed-first-layer kernel instead of the split-bf16 one
      env["OIVAE_ENC"] = "fp32"
    if mode == "onewave":
      env["OIVAE_DEC"] = "x3"   # the split-bf16 decoder with one wave per SIMD instead of the role-split pair
    if mode == "pair":
      env["OIVAE_DEC"] = "p"    # the role-split pair with both waves on bf16 terms (default: dec_x3_64g, both waves on fp16 terms, two groups per workgroup)
    if mode == "bf16":
      env["OIVAE_ENC"] = "bf16"  # bf16 encoder kernels; the decoder then has no ranges and takes the all-bf16 pair
    out = tmp_path / (mode + ".npz")
    subprocess.run([sys.executable, str(script), mode, str(out)], check=True, env=env, timeout=300)
    res[mode] = np.load(out)
  eng = _engine(24, 64, 64, 64, 64, seed=1)  # segment-wise comparison relative to each tensor's scale
  for other in ("generic", "layerwise", "fp32", "onewave", "pair", "bf16"):
    a, b = res["mfma"], res[other]
    assert np.max(np.abs(a["sc"][:6] - b["sc"][:6]) / np.maximum(np.abs(b["sc"][:6]), 1)) < 1e-5, other
    assert np.array_equal(a["z"], b["z"]) or np.max(np.abs(a["z"] - b["z"])) < 1e-5
    for name in eng.seg_off:
      o, n = eng.seg_off[name], eng.seg_numel[name]
      if n == 0:
        continue
      ga, gb = a["grads"][o:o + n], b["grads"][o:o + n]
      err = np.max(np.abs(ga - gb)) / max(np.max(np.abs(gb)), 1e-30)
      assert err < 2e-5, (other, name, err)


@pytest.mark.parametrize("K,H,M,dg,G", [(k, h, m, d, 7) for k in (32, 64) for h in (32, 64) for m in (32, 64) for d in (32, 64)
                                        if (k, h, m, d) not in ((64, 64, 64, 64), (32, 32, 32, 32))] +
                                       [(32, 32, 32, 32, 8), (64, 32, 32, 32, 8), (32, 32, 32, 32, 2), (32, 32, 32, 32, 7),
                                        (10, 8, 24, 64, 5), (16, 16, 48, 40, 6), (64, 64, 64, 48, 3), (33, 17, 64, 12, 4),
                                        (48, 48, 48, 48, 7), (1, 64, 40, 4, 9)])
def test_mixed_32_64_generators_on_the_matrix_decoder(K, H, M, dg, G):
  """Two-layer generators whose four dimensions are 32 or 64 in any combination take the matrix decoders (dz
  transposed) and run the all-64 fp16-term kernel (csrc/dec_pad64.hip): on zero-padded copies with 64 outputs per
  group; with 32 outputs on copies whose second layer is padded to 64 outputs while the kernel reads X by halves
  (dec_x3_64s<true>) -- except H = M = D_g = 32 with an even number of groups, whose neighbouring groups are PAIRED
  into all-64 virtual groups (the G = 8 and G = 2 cases; G = 7 is padded).  The shape-generic kernels on the same
  inputs are the check -- and dec_mfma<K, H, M, DG> on the fp32 matrix instruction (OIVAE_DEC=v1) as a third family:
  ragged row count, two samples, scalars 1e-5, every gradient tensor 2e-5 of its scale.  The last cases: widths
  OUTSIDE {32, 64} (any K, H, M up to 64, any uniform D_g up to 64 that is a multiple of 4) run the same padded
  launches; dec_mfma has no instance for them, the shape-generic kernels are the one check."""
  B, S = 150, 2   # 300 rows = 9.4 wave tiles
  lib = _lib.load()
  spec_eng = _engine(G, dg, K, H, M, seed=11)
  import ctypes
  assert lib.oivae_dz_layout(ctypes.byref(spec_eng._shape), B, S) == 1   # the matrix decoder's layout: this shape takes it
  g = torch.Generator(device="cuda").manual_seed(6)
  X = torch.rand(B, G * dg, generator=g, device="cuda")
  eps = torch.randn(S, B, K, generator=g, device="cuda")
  res = []
  families = [{}, {"OIVAE_FORCE_GENERIC": "1"}, {"OIVAE_DEC": "v1"}]
  if any(v not in (32, 64) for v in (K, H, M, dg)):
    families = families[:2]
  try:
    for env in families:
      _lib.set_test_hooks(**env)
      eng = _engine(G, dg, K, H, M, seed=11)
      eng.flat_grads.fill_(float("nan"))      # every gradient element must be written (the un-padding copies blocks)
      out = eng.forward_backward(X, eps, 0.7)
      torch.cuda.synchronize()
      res.append((eng.flat_grads.clone(), out["scalars"].clone(), eng))
      _lib.set_test_hooks(**{k: None for k in env})
  finally:
    _lib.set_test_hooks(OIVAE_FORCE_GENERIC=None, OIVAE_DEC=None)
  ga, sa, eng = res[0]
  for name in eng.seg_off:   # (the alignment gaps between the segments keep the fill)
    assert torch.isfinite(ga[eng.seg_off[name]:eng.seg_off[name] + eng.seg_numel[name]]).all(), name
  for gb, sb, _ in res[1:]:
    assert ((sa[:6] - sb[:6]).abs() / sb[:6].abs().clamp(min=1)).max().item() < 1e-5
    for name in eng.seg_off:
      o, n = eng.seg_off[name], eng.seg_numel[name]
      if n == 0:
        continue
      err = ((ga[o:o + n] - gb[o:o + n]).abs().max() / gb[o:o + n].abs().max().clamp(min=1e-30)).item()
      assert err < 2e-5, (name, err)


def test_full_scale_batch_equals_mean_of_half_batches():
  G, dg, K = 1024, 64, 64
  B = 65536
  eng = _engine(G, dg, K, 64, 64, seed=2)
  g = torch.Generator(device="cuda").manual_seed(9)
  X = torch.rand(B, G * dg, generator=g, device="cuda")
  eps = torch.randn(1, B, K, generator=g, device="cuda")
  out = eng.forward_backward(X, eps, 1.0)
  full_g = eng.flat_grads.clone()
  full_s = out["scalars"].clone()
  assert torch.isfinite(full_g).all() and torch.isfinite(full_s).all()
  acc_g = torch.zeros_like(full_g)
  acc_s = torch.zeros_like(full_s)
  h = B // 2
  for lo in (0, h):
    o = eng.forward_backward(X[lo:lo + h], eps[:, lo:lo + h], 1.0)
    acc_g += eng.flat_grads
    acc_s += o["scalars"]
  acc_g *= 0.5
  acc_s *= 0.5
  torch.cuda.synchronize()
  s_err = ((acc_s[:6] - full_s[:6]).abs() / full_s[:6].abs().clamp(min=1)).max().item()
  assert s_err < 1e-5, s_err
  for name in eng.seg_off:
    o, n = eng.seg_off[name], eng.seg_numel[name]
    if n == 0:
      continue
    a, b = acc_g[o:o + n], full_g[o:o + n]
    err = ((a - b).abs().max() / b.abs().max().clamp(min=1e-30)).item()
    assert err < 2e-5, (name, err)


def test_wide_prox_full_size_bit_exact_and_idempotent_zero_set():
  import ctypes
  from oi_vae_amd import _lib
  from oracle import prox_oracle
  L = _lib.init_device(0)
  G, K, H = 8192, 256, 8
  g = torch.Generator(device="cuda").manual_seed(4)
  W = torch.randn(G, K, H, generator=g, device="cuda")
  t = float(torch.sqrt((W.double() ** 2).sum(-1)).median())
  W0 = W.cpu().numpy()
  mask = torch.zeros(G, K, dtype=torch.uint8, device="cuda")
  st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  _lib.check(L.oivae_prox_f32(W.data_ptr(), G * K, H, t, mask.data_ptr(), None, st), "prox")
  torch.cuda.synchronize()
  want, wmask, _ = prox_oracle.proximal_step(W0, t)
  assert np.array_equal(W.cpu().numpy().view(np.uint32), want.view(np.uint32))
  assert np.array_equal(mask.cpu().numpy(), wmask)
  assert 0.45 < wmask.mean() < 0.55
  # a second, tiny threshold keeps every zero row zero; it may only kill rows whose norm
  # the first step left below 1e-6
  mask2 = torch.zeros_like(mask)
  _lib.check(L.oivae_prox_f32(W.data_ptr(), G * K, H, 1e-6, mask2.data_ptr(), None, st), "prox")
  torch.cuda.synchronize()
  m2 = mask2.cpu().numpy()
  assert np.all(m2 >= wmask)
  newly = (m2 == 1) & (wmask == 0)
  left = np.sqrt((want.astype(np.float64) ** 2).sum(-1))
  assert np.all(left[newly] <= 1.1e-6) and np.all(left[(m2 == 0)] > 0.9e-6)


@pytest.mark.parametrize("H,sgd", [(8, True), (16, False), (16, True), (32, False), (32, True)])
def test_persistent_prox_kernel_bit_exact_with_and_without_sgd(H, sgd):
  """Launches of 524288 rows and more with H in {8, 16, 32} take the persistent kernel that keeps the rsqrt14 table
  in LDS (prox.hip, path C): a ragged row count (the last pass is partial, and so is the last pair's workgroup)
  against the C oracle, bit for bit, with norms and mask; the same rows in one small launch each side of the
  threshold (path B) must give the same bits."""
  import ctypes
  from oi_vae_amd import _lib
  from oracle import prox_oracle
  L = _lib.init_device(0)
  rows = 4 * 256 * 512 + 4099
  g = torch.Generator(device="cuda").manual_seed(40 + H)
  W = torch.randn(rows, H, generator=g, device="cuda") * torch.rand(rows, 1, generator=g, device="cuda")
  dW = torch.randn(rows, H, generator=g, device="cuda")
  W[5].zero_(); W[rows - 1] *= 1e-20
  lr = 0.25
  t = float(torch.sqrt((W.double() ** 2).sum(-1)).median())
  # torch.optim.SGD on CPU rounds w + (-lr) g once (fma): in float64 the sum is exact before that one rounding
  start = (W.double() - lr * dW.double()).float() if sgd else W
  want, wmask, wnorm = prox_oracle.proximal_step(start.cpu().numpy().reshape(1, rows, H), t)
  st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

  def run(Wt, n):
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda")
    norms = torch.zeros(n, dtype=torch.float32, device="cuda")
    if sgd:
      _lib.check(L.oivae_sgd_prox_f32(Wt.data_ptr(), dW.data_ptr(), n, H, lr, t, 1, mask.data_ptr(), norms.data_ptr(), st), "sgd_prox")
    else:
      _lib.check(L.oivae_prox_f32(Wt.data_ptr(), n, H, t, mask.data_ptr(), norms.data_ptr(), st), "prox")
    torch.cuda.synchronize()
    return Wt.cpu().numpy(), mask.cpu().numpy(), norms.cpu().numpy()

  got, gmask, gnorm = run(W.clone(), rows)
  assert np.array_equal(got.view(np.uint32), want.reshape(rows, H).view(np.uint32))
  assert np.array_equal(gmask, wmask.reshape(rows))
  assert np.array_equal(gnorm.view(np.uint32), wnorm.reshape(rows).astype(np.float32).view(np.uint32))
  assert 0.4 < gmask.mean() < 0.6
  small = 100000   # below the threshold: the two-round-trip kernel
  got_s, gmask_s, gnorm_s = run(W[:small].clone(), small)
  assert np.array_equal(got_s.view(np.uint32), got[:small].view(np.uint32))
  assert np.array_equal(gmask_s, gmask[:small]) and np.array_equal(gnorm_s.view(np.uint32), gnorm[:small].view(np.uint32))


_ENC = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from oi_vae_amd import Engine, ModelSpec
dims = [37, 150, 91, 6, 64, 212]           # ragged groups, D = 560 (multiple of 4, not of 32)
D, K = sum(dims), 10                       # 2K = 20: a partial column tile
eng = Engine(ModelSpec(D=D, K=K, H=1, M=0, group_dims=dims, gen_type="linear"), device="cuda:0")
g = torch.Generator(device="cuda").manual_seed(2)
for n in ("enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig", "gen_W2", "gen_b2"):
  eng.p[n].copy_((torch.rand(eng.p[n].shape, generator=g, device="cuda") * 2 - 1) / 8)
eng.p["log_sigma"].fill_(float(np.log(1e-2)))
eng.p["Ws"].copy_(torch.randn(eng.p["Ws"].shape, generator=g, device="cuda"))
B = 777                                     # ragged row tiles too
X = torch.rand(B, D, generator=g, device="cuda")
eps = torch.randn(2, B, K, generator=g, device="cuda")
out = eng.forward_backward(X, eps, 0.7)
torch.cuda.synchronize()
np.savez(sys.argv[1], grads=eng.flat_grads.cpu().numpy(), sc=out["scalars"].cpu().numpy(), mu=out["mu_q"].cpu().numpy(),
         off=np.array([eng.seg_off[n] for n in ("enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig")]), n=np.array([eng.seg_numel["enc_Wmu"], K]))
'''


def test_mfma_encoder_on_ragged_shapes_agrees_with_generic(tmp_path):
  """MEG-like shape (K=10, D=560, B=777): the MFMA encoder's masked partial tiles against the generic kernels."""
  script = tmp_path / "enc.py"
  script.write_text(_ENC % {"root": ROOT})
  res = {}
  for mode in ("mfma", "generic"):
    env = dict(os.environ)
    if mode == "generic":
      env["OIVAE_FORCE_GENERIC"] = "1"
    out = tmp_path / (mode + ".npz")
    subprocess.run([sys.executable, str(script), str(out)], check=True, env=env, timeout=300)
    res[mode] = np.load(out)
  a, b = res["mfma"], res["generic"]
  assert np.max(np.abs(a["mu"] - b["mu"])) <= 1e-5 * max(1.0, np.max(np.abs(b["mu"])))
  assert np.max(np.abs(a["sc"][:6] - b["sc"][:6]) / np.maximum(np.abs(b["sc"][:6]), 1)) < 1e-5
  nW, K = int(a["n"][0]), int(a["n"][1])
  for o, n in zip(a["off"], (nW, K, nW, K)):
    ga, gb = a["grads"][o:o + n], b["grads"][o:o + n]
    assert np.max(np.abs(ga - gb)) <= 2e-5 * max(np.max(np.abs(gb)), 1e-30)


_FLAT = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from oi_vae_amd import Engine, ModelSpec
typ, H, dg = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
G, K = 200, 48                                  # G not a multiple of the 16-group blocks, K a partial 32-tile
eng = Engine(ModelSpec(D=G * dg, K=K, H=H, M=0, group_dims=[dg] * G, gen_type=typ), device="cuda:0")
g = torch.Generator(device="cuda").manual_seed(4)
for n in ("enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig", "gen_W2", "gen_b2"):
  scale = 0.25 if n.startswith("gen") else 0.01   # z of order one
  eng.p[n].copy_((torch.rand(eng.p[n].shape, generator=g, device="cuda") * 2 - 1) * scale)
eng.p["log_sigma"].copy_(torch.log(torch.rand(eng.p["log_sigma"].shape, generator=g, device="cuda") * 0.05 + 0.01))
# small Ws keeps tanh out of saturation, where 1 - t^2 cancels and ANY fp32 evaluation order is only 1e-4 accurate
eng.p["Ws"].copy_(0.15 * torch.randn(eng.p["Ws"].shape, generator=g, device="cuda"))
B, S = 150, 3                                   # ragged row tiles, several Monte-Carlo samples
X = torch.rand(B, G * dg, generator=g, device="cuda")
eps = torch.randn(S, B, K, generator=g, device="cuda")
out = eng.forward_backward(X, eps, 0.7)
torch.cuda.synchronize()
np.savez(sys.argv[1], grads=eng.flat_grads.cpu().numpy(), sc=out["scalars"].cpu().numpy())
'''


@pytest.mark.parametrize("typ,H,dg", [("linear", 8, 4), ("tanh_linear", 8, 8), ("tanh_linear", 4, 4), ("linear", 4, 8)])
def test_flat_many_small_groups_decoder_agrees_with_generic(tmp_path, typ, H, dg):
  """The GEMM-decomposed decoder for many small groups (wide-prox shape family) against the generic kernels."""
  script = tmp_path / "flat.py"
  script.write_text(_FLAT % {"root": ROOT})
  res = {}
  for mode in ("flat", "generic"):
    env = dict(os.environ)
    if mode == "generic":
      env["OIVAE_FORCE_GENERIC"] = "1"
    out = tmp_path / (mode + ".npz")
    subprocess.run([sys.executable, str(script), str(out), typ, str(H), str(dg)], check=True, env=env, timeout=300)
    res[mode] = np.load(out)
  a, b = res["flat"], res["generic"]
  assert np.max(np.abs(a["sc"][:6] - b["sc"][:6]) / np.maximum(np.abs(b["sc"][:6]), 1)) < 1e-5
  from oi_vae_amd import ModelSpec
  from oi_vae_amd.engine import _SEG_ORDER, _ALIGN
  shapes = ModelSpec(D=200 * dg, K=48, H=H, M=0, group_dims=[dg] * 200, gen_type=typ).seg_shapes()
  off = 0
  for name in _SEG_ORDER:
    n = int(np.prod(shapes[name]))
    if n:
      ga, gb = a["grads"][off:off + n], b["grads"][off:off + n]
      err = np.max(np.abs(ga - gb)) / max(np.max(np.abs(gb)), 1e-30)
      assert err < 2e-5, (name, err)
    off += (n + _ALIGN - 1) // _ALIGN * _ALIGN


def test_full_wide_prox_shape_whole_step():
  """BASELINE.json configs[4] at its full size -- G = 8192 groups of 4 outputs, K = 256, H = 8, B = 1024, single-layer
  generators -- through the whole step: the flat (GEMM-decomposed) decoder against the shape-generic kernels on the
  same inputs (scalars 1e-5, gradients 2e-5 of scale), then Adam and the fused SGD + proximal step with the
  threshold at the median row norm: about half of the 2 M rows die and the result equals the C oracle's on the
  same pre-step bits, bit for bit."""
  from oi_vae_amd import Engine, ModelSpec
  from oracle import prox_oracle
  G, dg, K, H, B = 8192, 4, 256, 8, 1024
  eng = Engine(ModelSpec(D=G * dg, K=K, H=H, M=0, group_dims=[dg] * G, gen_type="linear"), device="cuda:0")
  g = torch.Generator(device="cuda").manual_seed(12)
  for n in ("enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig", "gen_W2", "gen_b2"):
    scale = 0.25 if n.startswith("gen") else 0.003
    eng.p[n].copy_((torch.rand(eng.p[n].shape, generator=g, device="cuda") * 2 - 1) * scale)
  eng.p["log_sigma"].fill_(float(np.log(1e-2)))
  eng.p["Ws"].copy_(torch.randn(eng.p["Ws"].shape, generator=g, device="cuda"))
  X = torch.rand(B, G * dg, generator=g, device="cuda")
  eps = torch.randn(1, B, K, generator=g, device="cuda")
  res = {}
  try:
    for mode in ("generic", "flat"):
      _lib.set_test_hooks(OIVAE_FORCE_GENERIC="1" if mode == "generic" else None)
      eng._ws_key = None
      out = eng.forward_backward(X, eps, 1.0)
      torch.cuda.synchronize()
      res[mode] = (eng.flat_grads.double().cpu().clone(), out["scalars"].double().cpu().clone())
  finally:
    _lib.set_test_hooks(OIVAE_FORCE_GENERIC=None)
  a, b = res["flat"], res["generic"]
  assert torch.isfinite(a[0]).all()
  assert float(((a[1][:6] - b[1][:6]).abs() / b[1][:6].abs().clamp(min=1.0)).max()) < 1e-5
  for name in eng.seg_off:
    o, n = eng.seg_off[name], eng.seg_numel[name]
    if n:
      ga, gb = a[0][o:o + n], b[0][o:o + n]
      assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max()) + 1e-30, name
  # optimizers + proximal step on the flat path's gradients
  W0 = eng.p["Ws"].cpu().numpy().copy()
  dW = eng.g["Ws"].cpu().numpy().copy()
  lr = 1e-4
  t = float(torch.linalg.vector_norm(eng.p["Ws"], dim=2).median())
  eng.adam_step(1e-5)   # (32768 inputs moving by 1e-3 each in step would saturate the encoder's sigma head: a dead gradient)
  eng.sgd_prox_step(lr, t)
  torch.cuda.synchronize()
  pre = (W0.astype(np.float64) + np.float64(np.float32(-lr)) * dW.astype(np.float64)).astype(np.float32)
  want, wmask, _ = prox_oracle.proximal_step(pre, t)
  assert np.array_equal(eng.p["Ws"].cpu().numpy().view(np.uint32), want.view(np.uint32))
  assert np.array_equal(eng.zero_mask.cpu().numpy(), wmask)
  assert 0.3 < wmask.mean() < 0.7
  # second step, on the sparse state the proximal step left (half the rows of Ws exactly zero), with a few groups dead
  # altogether, sigmas spread over two decades (the gradients of the pre-activations then span as much) and another
  # batch whose maximum lies in another binade: the flat decoder on two fp16 terms (default), the flat decoder on the
  # fp32 matrix instruction (OIVAE_ENC=fp32) and the shape-generic kernels must still agree
  eng.p["Ws"][:5].zero_()
  eng.p["log_sigma"].copy_(torch.log(10 ** (torch.rand(G * dg, generator=g, device="cuda") * 2 - 2.5)))
  X2 = 3.0 * torch.rand(B, G * dg, generator=g, device="cuda")
  res = {}
  try:
    for mode, hooks in (("generic", dict(OIVAE_FORCE_GENERIC="1")), ("flat_h2", {}), ("flat_fp32", dict(OIVAE_ENC="fp32"))):
      _lib.set_test_hooks(OIVAE_FORCE_GENERIC=hooks.get("OIVAE_FORCE_GENERIC"), OIVAE_ENC=hooks.get("OIVAE_ENC"))
      eng._ws_key = None
      out = eng.forward_backward(X2, eps, 1.0)
      torch.cuda.synchronize()
      res[mode] = (eng.flat_grads.double().cpu().clone(), out["scalars"].double().cpu().clone())
  finally:
    _lib.set_test_hooks(OIVAE_FORCE_GENERIC=None, OIVAE_ENC=None)
    eng._ws_key = None
  b = res["generic"]
  for mode in ("flat_h2", "flat_fp32"):
    a = res[mode]
    assert torch.isfinite(a[0]).all(), mode
    assert float(((a[1][:6] - b[1][:6]).abs() / b[1][:6].abs().clamp(min=1.0)).max()) < 1e-5, mode
    for name in eng.seg_off:
      o, n = eng.seg_off[name], eng.seg_numel[name]
      if n:
        ga, gb = a[0][o:o + n], b[0][o:o + n]
        assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max()) + 1e-30, (mode, name)


@pytest.mark.parametrize("typ,H,K", [("linear", 1, 10), ("tanh_linear", 8, 8), ("linear", 2, 16), ("tanh_linear", 4, 33),
                                     ("tanh_linear", 1, 10)])
def test_single_layer_decoder_agrees_with_generic(typ, H, K):
  """dec_single (single-layer generators, skinny H: the bars / mocap / MEG scripts' own shapes) against dec_generic on
  MEG-like ragged groups: widths from 1 to 300 (several 64-column chunks, partial chunks, one-column groups), a row
  count that ends in a partial 64-row tile, three Monte-Carlo samples (rows of X repeat), odd K.  Scalars 1e-5,
  gradients 2e-5 of scale; the weight gradients of dec_single must also be bit-reproducible run to run."""
  from oi_vae_amd import Engine, ModelSpec
  rng = np.random.default_rng(5)
  dims = [1, 300, 64, 65, 7, 128, 129, 2, 63, 191] + rng.integers(20, 260, 12).tolist()
  dims[-1] += (-sum(dims)) % 4
  G, D, B, S = len(dims), sum(dims), 333, 3
  eng = Engine(ModelSpec(D=D, K=K, H=H, M=0, group_dims=dims, gen_type=typ), device="cuda:0")
  g = torch.Generator(device="cuda").manual_seed(9)
  for n in ("enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig", "gen_W2", "gen_b2"):
    scale = 0.3 if n.startswith("gen") else 0.02
    eng.p[n].copy_((torch.rand(eng.p[n].shape, generator=g, device="cuda") * 2 - 1) * scale)
  eng.p["log_sigma"].copy_(torch.log(torch.rand(eng.p["log_sigma"].shape, generator=g, device="cuda") * 0.3 + 0.02))
  eng.p["Ws"].copy_(0.4 * torch.randn(eng.p["Ws"].shape, generator=g, device="cuda"))
  X = torch.rand(B, D, generator=g, device="cuda")
  eps = torch.randn(S, B, K, generator=g, device="cuda")
  res = {}
  try:
    for mode in ("generic", "single", "single_again"):
      _lib.set_test_hooks(OIVAE_FORCE_GENERIC="1" if mode == "generic" else None)
      eng._ws_key = None
      out = eng.forward_backward(X, eps, 0.8)
      torch.cuda.synchronize()
      res[mode] = (eng.flat_grads.double().cpu().clone(), out["scalars"].double().cpu().clone())
  finally:
    _lib.set_test_hooks(OIVAE_FORCE_GENERIC=None)
  a, b = res["single"], res["generic"]
  assert torch.isfinite(a[0]).all()
  assert float(((a[1][:6] - b[1][:6]).abs() / b[1][:6].abs().clamp(min=1.0)).max()) < 1e-5
  for name in eng.seg_off:
    o, n = eng.seg_off[name], eng.seg_numel[name]
    if n:
      ga, gb = a[0][o:o + n], b[0][o:o + n]
      assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max()) + 1e-30, name
  for name in ("gen_W2", "gen_b2", "log_sigma", "Ws"):     # fixed summation order: no float atomics behind these
    o, n = eng.seg_off[name], eng.seg_numel[name]
    assert torch.equal(res["single"][0][o:o + n], res["single_again"][0][o:o + n]), name
