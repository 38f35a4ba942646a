"""GPU tests of the encoder GEMMs that carry every fp32 operand in 16-bit terms (oi_vae_amd/csrc/x3_encoder.hip): the
default, two scaled fp16 terms and three fp16 matrix products per fp32 product ("x3" below, the library's default
mode), and three bf16 terms with six products (OIVAE_ENC=bf16).  The claim under test: the result is fp32-grade --
against a float64 product of the same fp32 inputs its error is no larger than that of the fp32 matrix
instructions (OIVAE_ENC=fp32 selects those; `_lib.set_test_hooks` makes the library read the variable again).
Tolerances: 5e-6 of the result's largest magnitude absolute, and at most 2x the fp32 path's own error + 2e-6."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _hooks(**env):
  from oi_vae_amd import _lib
  _lib.set_test_hooks(**env)


def _mode(name):
  _hooks(OIVAE_ENC=name if name in ("fp32", "bf16") else None)


@pytest.fixture(autouse=True)
def _restore_env():
  yield
  _hooks(OIVAE_ENC=None, OIVAE_DEC=None)


def _engine(D, K, seed):
  from oi_vae_amd import Engine, ModelSpec
  G = D // 64
  eng = Engine(ModelSpec(D=D, K=K, H=64, M=64, group_dims=[64] * G, gen_type="mlp"), device="cuda:0")
  g = torch.Generator(device="cuda").manual_seed(seed)
  for n in ("enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig"):
    eng.p[n].copy_((torch.rand(eng.p[n].shape, generator=g, device="cuda") * 2 - 1) / D ** 0.5)
  return eng, g


def _encoder_fwd(eng, X, eps, fresh_ws=True):
  from oi_vae_amd import _lib
  B, K = X.shape[0], eng.spec.K
  if fresh_ws:
    eng._ws_key = None
  ws = eng._workspace(B, eps.shape[0])
  mu, sig, ea = (torch.empty(B, K, device="cuda") for _ in range(3))
  z = torch.empty(eps.shape[0] * B, K, device="cuda")
  pm, ps = torch.zeros(1, K, device="cuda"), torch.ones(1, K, device="cuda")
  _lib.check(eng.lib.oivae_encoder_fwd_f32(ctypes.byref(eng._shape), ctypes.byref(eng._pt), X.data_ptr(), B, eps.shape[0],
             eps.data_ptr(), 1.0, 1e-3, pm.data_ptr(), ps.data_ptr(), 1, mu.data_ptr(), sig.data_ptr(), ea.data_ptr(),
             z.data_ptr(), ws.data_ptr(), eng._ws_bytes, eng._stream()), "encoder_fwd")
  torch.cuda.synchronize()
  return mu.double().cpu()


def _check(err, scale):
  for m in err:
    if m != "fp32":
      assert err[m] <= 5e-6 * scale, err
      # (the fp32 kernels sum some shapes in slices -- K = 32 at 4096 rows comes out at 6e-7 of scale -- a moving bar)
      assert err[m] <= 2.0 * err["fp32"] + 2e-6 * scale, err


@pytest.mark.parametrize("B,D,K", [(4096, 8192, 64), (200, 1536, 64), (1000, 4096, 64), (1000, 4096, 32), (2048, 2048, 96)])
def test_forward_product_is_fp32_grade(B, D, K):
  """(4096, 8192): whole 128-row tiles; (200, 1536) and (1000, 4096): ragged last row tile, split-K images; K = 32 and
  96: 2K = 64 and 192, half-used column tiles of the fp16 kernels (zero rows in the weight image)."""
  eng, g = _engine(D, K, seed=3)
  X = torch.rand(B, D, generator=g, device="cuda")
  eps = torch.zeros(1, B, K, device="cuda")
  want = (X.double() @ eng.p["enc_Wmu"].double().t() + eng.p["enc_bmu"].double()).cpu()
  err = {}
  for mode in ("fp32", "bf16", "x3"):
    _mode(mode)
    err[mode] = float((_encoder_fwd(eng, X, eps) - want).abs().max())
  _check(err, float(want.abs().max()))


def test_forward_wide_dynamic_range_and_signs():
  """Operands from 1e-30 to 1e+12 with both signs, exact zeros and denormals in one product.  bf16 keeps fp32's
  exponent range and needs no scaling; the fp16 kernel scales by the power of two of max|X| and keeps every element
  to an absolute error of 2^-39 of that maximum, which is what a product judged at the scale of its largest terms
  needs (every row here has terms near 1e+12)."""
  B, D = 512, 2048
  eng, g = _engine(D, 64, seed=5)
  mag = torch.rand(B, D, generator=g, device="cuda") * 42 - 30          # exponents -30 .. +12
  X = torch.pow(10.0, mag) * torch.sign(torch.rand(B, D, generator=g, device="cuda") - 0.5)
  X[:, ::7] = 0.0
  X[:, 3::11] = 1e-41                                                     # fp32 denormals
  eps = torch.zeros(1, B, 64, device="cuda")
  want = (X.double() @ eng.p["enc_Wmu"].double().t() + eng.p["enc_bmu"].double()).cpu()
  err = {}
  for mode in ("fp32", "bf16", "x3"):
    _mode(mode)
    got = _encoder_fwd(eng, X, eps)
    assert torch.isfinite(got).all()
    # row-wise scale: every row has its own largest term
    err[mode] = float(((got - want).abs().max(dim=1).values / want.abs().max(dim=1).values).max())
  _check(err, 1.0)


def test_backward_weight_gradient_is_fp32_grade():
  """oivae_encoder_bwd_f32 with mu_q = 0, sig_q = 1, eps = 0 under a standard-normal prior makes the
  pre-activation gradient [dz, 0]: dWmu = dz^T X is checked against float64, dWsig must be exactly zero."""
  from oi_vae_amd import _lib
  B, D, K = 4096, 8192, 64
  eng, g = _engine(D, K, seed=7)
  X = torch.rand(B, D, generator=g, device="cuda")
  dzT = torch.randn(K, B, generator=g, device="cuda") / B               # the MFMA decoder hands dz over as [K, N]
  zeros, ones = torch.zeros(B, K, device="cuda"), torch.ones(B, K, device="cuda")
  eps = torch.zeros(1, B, K, device="cuda")
  pm, ps = torch.zeros(1, K, device="cuda"), torch.ones(1, K, device="cuda")
  want = (dzT.double() @ X.double()).cpu()
  err = {}
  for mode in ("fp32", "x3"):
    _mode(mode)
    eng._ws_key = None
    ws = eng._workspace(B, 1)
    eng.flat_grads.fill_(float("nan"))                                    # every element must be written
    _lib.check(eng.lib.oivae_encoder_bwd_f32(ctypes.byref(eng._shape), ctypes.byref(eng._gt), X.data_ptr(), B, 1,
               eps.data_ptr(), 1.0, pm.data_ptr(), ps.data_ptr(), 1, zeros.data_ptr(), ones.data_ptr(), ones.data_ptr(),
               dzT.data_ptr(), ws.data_ptr(), eng._ws_bytes, eng._stream()), "encoder_bwd")
    torch.cuda.synchronize()
    err[mode] = float((eng.g["enc_Wmu"].double().cpu() - want).abs().max())
    assert float(eng.g["enc_Wsig"].abs().max()) == 0.0
    bias = eng.g["enc_bmu"].double().cpu()
    assert float((bias - dzT.double().sum(dim=1).cpu()).abs().max()) <= 1e-6 * float(bias.abs().max())
  _check(err, float(want.abs().max()))


def test_x3_path_is_deterministic():
  eng, g = _engine(4096, 64, seed=9)
  X = torch.rand(640, 4096, generator=g, device="cuda")
  eps = torch.zeros(1, 640, 64, device="cuda")
  a = _encoder_fwd(eng, X, eps)
  b = _encoder_fwd(eng, X, eps)
  assert torch.equal(a, b)


def test_decoder_gradients_are_fp32_grade():
  """oivae_decoder_fwd_bwd_f32 at the all-64 generator shape (dec_x3_64s by default, dec_x3_64g under OIVAE_DEC=g, dec_x3_64 under
  OIVAE_DEC=x3, dec_mfma64_v4 under OIVAE_DEC=v4)
  against float64 autograd of -loglik / (S B) on the same fp32 inputs: ragged row count, two samples per row."""
  from oi_vae_amd import Engine, ModelSpec, _lib
  G, K, B, S = 6, 64, 150, 2
  D, N = G * 64, B * S
  eng = Engine(ModelSpec(D=D, K=K, H=64, M=64, group_dims=[64] * G, gen_type="mlp"), device="cuda:0")
  g = torch.Generator(device="cuda").manual_seed(21)
  for n in ("gen_W1", "gen_b1", "gen_W2", "gen_b2"):
    eng.p[n].copy_((torch.rand(eng.p[n].shape, generator=g, device="cuda") * 2 - 1) / 8)
  eng.p["log_sigma"].copy_(torch.rand(eng.p["log_sigma"].shape, generator=g, device="cuda") * 0.5 - 1.5)
  eng.p["Ws"].copy_(torch.randn(eng.p["Ws"].shape, generator=g, device="cuda") * 0.3)
  X = torch.rand(B, D, generator=g, device="cuda")
  z = torch.randn(N, K, generator=g, device="cuda")

  # float64 reference (lib/models.py:91-95, lib/distributions.py:40-45, lib/oivae.py:55-59)
  p64 = {n: eng.p[n].double().clone().requires_grad_(True) for n in ("gen_W1", "gen_b1", "gen_W2", "gen_b2", "log_sigma", "Ws")}
  z64 = z.double().clone().requires_grad_(True)
  W1 = p64["gen_W1"].view(G, 64, 64); b1 = p64["gen_b1"].view(G, 64)
  W2 = p64["gen_W2"].view(G, 64, 64); b2 = p64["gen_b2"].view(G, 64)
  h = torch.einsum("nk,gkh->gnh", z64, p64["Ws"].view(G, K, 64))
  a1 = torch.einsum("gnh,gmh->gnm", h, W1) + b1[:, None, :]
  mu = torch.einsum("gnm,gdm->gnd", torch.tanh(a1), W2) + b2[:, None, :]
  sig = torch.exp(p64["log_sigma"].view(G, 64)) + 1e-3
  x = X.double().repeat(S, 1).view(N, G, 64).permute(1, 0, 2)
  ll = (-0.5 * np.log(2 * np.pi) - torch.log(sig)[:, None, :] - 0.5 * ((x - mu) / sig[:, None, :]) ** 2).sum()
  (-ll / (S * B)).backward()
  want = {n: p64[n].grad.cpu() for n in p64}
  want_dz = z64.grad.cpu()

  err = {}
  for mode in ("v4", "x3", "g", "default"):
    _hooks(OIVAE_DEC=None if mode == "default" else mode)
    eng._ws_key = None
    ws = eng._workspace(B, S)
    eng.flat_grads.fill_(float("nan"))
    dz = torch.empty(K, N, device="cuda")      # the matrix-core decoders hand dz over transposed
    _lib.check(eng.lib.oivae_decoder_fwd_bwd_f32(ctypes.byref(eng._shape), ctypes.byref(eng._pt), ctypes.byref(eng._gt),
               X.data_ptr(), B, S, z.data_ptr(), 1e-3, dz.data_ptr(), ws.data_ptr(), eng._ws_bytes, eng._stream()), "decoder")
    torch.cuda.synchronize()
    e = {"dz": float((dz.t().double().cpu() - want_dz).abs().max() / want_dz.abs().max())}
    for n in want:
      got = eng.g[n].double().cpu().view_as(want[n])
      e[n] = float((got - want[n]).abs().max() / want[n].abs().max())
    err[mode] = e
  _hooks(OIVAE_DEC=None)
  for mode in ("x3", "g", "default"):
    for n in err[mode]:
      assert err[mode][n] <= 5e-6, (mode, n, err)
      assert err[mode][n] <= 1.5 * err["v4"][n] + 3e-7, (mode, n, err)


@pytest.mark.parametrize("B,S", [(1, 1), (5, 1), (33, 3), (64, 1), (97, 2)])
def test_paired_decoder_tiny_batches(B, S):
  """dec_x3_64s (default) and dec_x3_64g at batches that leave tile streams empty (fewer than four 32-row tiles), end in a ragged tile or
  repeat rows over samples: same scalars and gradients as the fp32-instruction kernel dec_mfma64_v4."""
  from tests.test_gpu_scale import _engine
  G = 5
  eng = _engine(G, 64, 64, 64, 64, seed=11)
  g = torch.Generator(device="cuda").manual_seed(100 + B)
  X = torch.rand(B, G * 64, generator=g, device="cuda")
  eps = torch.randn(S, B, 64, generator=g, device="cuda")
  res = {}
  for mode in ("v4", "g", None):
    _hooks(OIVAE_DEC=mode)
    eng._ws_key = None
    out = eng.forward_backward(X, eps, 0.9)
    torch.cuda.synchronize()
    res[mode] = (eng.flat_grads.double().cpu().clone(), out["scalars"].double().cpu().clone())
  _hooks(OIVAE_DEC=None)
  for mode in (None, "g"):
    a, b = res[mode], res["v4"]
    assert torch.isfinite(a[0]).all()
    assert float(((a[1][:6] - b[1][:6]).abs() / b[1][:6].abs().clamp(min=1.0)).max()) < 1e-5
    for name in eng.seg_off:
      o, n = eng.seg_off[name], eng.seg_numel[name]
      if n:
        ga, gb = a[0][o:o + n], b[0][o:o + n]
        assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max()) + 1e-12, (mode, name)


def test_fp16_forward_depends_on_its_inputs_only():
  """The fp16 forward kernel starts from the scale of the previous call's X (kept in the workspace) and repeats the
  product when this call's maximum asks for another one: whatever was multiplied before -- larger, smaller, all-zero
  inputs, a fresh workspace -- the result for a given X is bit-identical and fp32-grade."""
  B, D = 640, 4096
  eng, g = _engine(D, 64, seed=9)
  X = torch.rand(B, D, generator=g, device="cuda")
  eps = torch.zeros(1, B, 64, device="cuda")
  want = (X.double() @ eng.p["enc_Wmu"].double().t() + eng.p["enc_bmu"].double()).cpu()
  first = _encoder_fwd(eng, X, eps)                                    # fresh workspace: no usable history
  assert float((first - want).abs().max()) <= 5e-6 * float(want.abs().max())
  again = _encoder_fwd(eng, X, eps, fresh_ws=False)                    # history = the same X: first launch accepted
  assert torch.equal(first, again)
  for factor in (4096.0, 1.0 / 65536.0, 0.0, 3.0e-20, 7.0e15):
    Y = X * factor
    wantY = (Y.double() @ eng.p["enc_Wmu"].double().t() + eng.p["enc_bmu"].double()).cpu()
    gotY = _encoder_fwd(eng, Y, eps, fresh_ws=False)
    assert torch.isfinite(gotY).all()
    assert float((gotY - wantY).abs().max()) <= 5e-6 * max(float(wantY.abs().max()), 1e-30), factor
    back = _encoder_fwd(eng, X, eps, fresh_ws=False)                   # ... and back to X after each of them
    assert torch.equal(first, back), factor


@pytest.mark.parametrize("K", [64, 32, 96])
def test_fp16_backward_weight_gradient_is_fp32_grade(K):
  """The fp16 backward kernel needs max|X| from the forward call of the same step (same X, batch and workspace);
  without that call the library takes the bf16 kernel (test_backward_weight_gradient_is_fp32_grade).  Here the forward
  call comes first: dWmu = dz^T X against float64 for the fp16, bf16 and fp32-instruction kernels, gradients of
  very different magnitudes included.  K = 32 and 96: column tiles half used (2K = 64, 192); the sigma head's rows of
  the gradient must still be written (zeros here)."""
  from oi_vae_amd import _lib
  B, D = 4096, 8192
  eng, g = _engine(D, K, seed=17)
  X = torch.rand(B, D, generator=g, device="cuda")
  zeros, ones = torch.zeros(B, K, device="cuda"), torch.ones(B, K, device="cuda")
  eps = torch.zeros(1, B, K, device="cuda")
  pm, ps = torch.zeros(1, K, device="cuda"), torch.ones(1, K, device="cuda")
  for gscale in (1.0, 1e-9, 3e4):
    dzT = torch.randn(K, B, generator=g, device="cuda") * (gscale / B)
    want = (dzT.double() @ X.double()).cpu()
    # the layout the decoder of this shape hands dz over in: [K, N] from the matrix-core decoders, else [N, K]
    dz_arg = dzT if eng.lib.oivae_dz_layout(ctypes.byref(eng._shape), B, 1) == 1 else dzT.t().contiguous()
    err = {}
    for mode in ("fp32", "bf16", "x3"):
      _mode(mode)
      _encoder_fwd(eng, X, eps)                                           # records max|X| for the backward call
      ws = eng._workspace(B, 1)
      eng.flat_grads.fill_(float("nan"))
      _lib.check(eng.lib.oivae_encoder_bwd_f32(ctypes.byref(eng._shape), ctypes.byref(eng._gt), X.data_ptr(), B, 1,
                 eps.data_ptr(), 1.0, pm.data_ptr(), ps.data_ptr(), 1, zeros.data_ptr(), ones.data_ptr(), ones.data_ptr(),
                 dz_arg.data_ptr(), ws.data_ptr(), eng._ws_bytes, eng._stream()), "encoder_bwd")
      torch.cuda.synchronize()
      got = eng.g["enc_Wmu"].double().cpu()
      assert torch.isfinite(got).all()
      err[mode] = float((got - want).abs().max())
      assert float(eng.g["enc_Wsig"].abs().max()) == 0.0
    _check(err, float(want.abs().max()))


def test_fp16_backward_scales_every_encoder_unit_by_its_own_gradient():
  """The image of dpre^T is scaled per column of dpre (absmax_cols_f32): an encoder unit whose gradient lies many
  orders below the others' -- a saturated sigma head -- keeps fp32-grade accuracy relative to ITS OWN row of the weight
  gradient.  Rows of dz from 1 down to 1e-30 (and one all-zero row) against float64, row by row."""
  from oi_vae_amd import _lib
  B, D, K = 2048, 4096, 64
  eng, g = _engine(D, K, seed=23)
  X = torch.rand(B, D, generator=g, device="cuda")
  zeros, ones = torch.zeros(B, K, device="cuda"), torch.ones(B, K, device="cuda")
  eps = torch.zeros(1, B, K, device="cuda")
  pm, ps = torch.zeros(1, K, device="cuda"), torch.ones(1, K, device="cuda")
  fac = 10.0 ** (-torch.arange(K, device="cuda", dtype=torch.float32) * (30.0 / (K - 1)))
  fac[7] = 0.0
  dzT = torch.randn(K, B, generator=g, device="cuda") * fac[:, None]
  want = (dzT.double() @ X.double()).cpu()
  _mode("x3")
  _encoder_fwd(eng, X, eps)
  ws = eng._workspace(B, 1)
  eng.flat_grads.fill_(float("nan"))
  _lib.check(eng.lib.oivae_encoder_bwd_f32(ctypes.byref(eng._shape), ctypes.byref(eng._gt), X.data_ptr(), B, 1,
             eps.data_ptr(), 1.0, pm.data_ptr(), ps.data_ptr(), 1, zeros.data_ptr(), ones.data_ptr(), ones.data_ptr(),
             dzT.data_ptr(), ws.data_ptr(), eng._ws_bytes, eng._stream()), "encoder_bwd")
  torch.cuda.synchronize()
  got = eng.g["enc_Wmu"].double().cpu()
  assert torch.isfinite(got).all()
  row_scale = want.abs().max(dim=1).values
  row_err = (got - want).abs().max(dim=1).values
  assert float(row_err[7]) == 0.0 and float(row_scale[7]) == 0.0
  live = row_scale > 0
  assert float((row_err[live] / row_scale[live]).max()) < 5e-6, (row_err / row_scale.clamp(min=1e-300)).tolist()


@pytest.mark.parametrize("case", ["plain", "mixed_sigma", "large_x", "small_latent", "big_weights", "odd_groups",
                                  "dead_group", "huge_w2_row", "x_span", "saturated_tanh", "x_zero"])
def test_fp16_chain_decoder_step_against_float64(case):
  """The whole step at the all-64 shape (fp16 encoder kernels, dec_x3_64g) against float64 autograd of the same loss
  on the same fp32 inputs, beside the fp32-instruction kernels (OIVAE_ENC=fp32 OIVAE_DEC=v4) whose error sets the
  bar.  The cases move the ranges the fp16 scales are derived from: per-dimension sigmas from 2e-3 to 3 (gmu's bound
  is max c3 times a bound of mu - x), inputs of magnitude 300, latents of magnitude 1e-3, generator weights of
  magnitude 4; an odd number of groups (the last workgroup of dec_x3_64g holds one group); states that training with
  the proximal step produces -- a group whose Ws is all zero (max|A| = 0: the scale-1 path) next to groups with most
  rows zeroed; a W2 with one row 10^4 times the others (every bound derived from row sums is loose by that much for
  the other rows); X spanning 2^20 inside every row; pre-activations deep in tanh's saturation; X = 0 (max|X| = 0)."""
  from tests.test_gpu_scale import _engine as scale_engine
  G, K, B, S = 6, 64, 150, 2
  if case == "odd_groups":
    G = 5
  eng = scale_engine(G, 64, K, 64, 64, seed=31)
  g = torch.Generator(device="cuda").manual_seed(77)
  xs = 1.0
  if case == "dead_group":
    eng.p["Ws"][2].zero_()                                   # a group the proximal step has killed
    dead = torch.rand(G, K, generator=g, device="cuda") < 0.7
    eng.p["Ws"][dead] = 0.0                                   # and 70 % of the other groups' rows
    eng.p["Ws"][4, :, :] *= 1e-4                              # one group almost dead: tiny A beside O(1) partners
  if case == "huge_w2_row":
    eng.p["gen_W2"].view(G, 64, 64)[:, 7, :] *= 1e4
    eng.p["gen_b2"].view(G, 64)[:, 7] = 50.0
    eng.p["log_sigma"].view(G, 64)[:, 7] = float(np.log(30.0))
  if case == "saturated_tanh":
    eng.p["gen_W1"].mul_(40.0)
  if case == "mixed_sigma":
    eng.p["log_sigma"].copy_(torch.log(torch.pow(10.0, torch.rand(eng.p["log_sigma"].shape, generator=g, device="cuda") * 3.2 - 2.7)))
  if case == "large_x":
    xs = 300.0
    for n in ("enc_Wmu", "enc_Wsig"):
      eng.p[n].mul_(1.0 / 300.0)     # keep the pre-activations (and exp of them) in range
  if case == "small_latent":
    for n in ("enc_Wmu", "enc_bmu"):
      eng.p[n].mul_(1e-3)
    eng.p["enc_bsig"].fill_(float(np.log(1e-3)))
  if case == "big_weights":
    eng.p["gen_W2"].mul_(32.0)
    eng.p["gen_W1"].mul_(4.0)
  X = torch.rand(B, G * 64, generator=g, device="cuda") * xs
  if case == "x_span":
    X = X * torch.pow(2.0, -20.0 * torch.rand(B, G * 64, generator=g, device="cuda"))
  if case == "x_zero":
    X.zero_()
    eng.p["enc_bmu"].normal_(generator=g)                     # z must not vanish with X
  eps = torch.randn(S, B, K, generator=g, device="cuda")

  # float64 reference of loss = KL - loglik/(S B) - theta prior (lib/oivae.py:42-79), gradients by autograd
  names = ("enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig", "gen_W1", "gen_b1", "gen_W2", "gen_b2", "log_sigma", "Ws")
  p64 = {n: eng.p[n].double().clone().requires_grad_(True) for n in names}
  X64 = X.double()
  mu_q = X64 @ p64["enc_Wmu"].view(K, -1).t() + p64["enc_bmu"]
  sig_q = torch.exp(X64 @ p64["enc_Wsig"].view(K, -1).t() + p64["enc_bsig"]) * 0.7 + 1e-3
  z = (mu_q[None] + sig_q[None] * eps.double()).reshape(S * B, K)
  kl = (-0.5 + (sig_q ** 2 + mu_q ** 2) / 2 - torch.log(sig_q)).sum() / B
  h = torch.einsum("nk,gkh->gnh", z, p64["Ws"].view(G, K, 64))
  a1 = torch.einsum("gnh,gmh->gnm", h, p64["gen_W1"].view(G, 64, 64)) + p64["gen_b1"].view(G, 1, 64)
  mu = torch.einsum("gnm,gdm->gnd", torch.tanh(a1), p64["gen_W2"].view(G, 64, 64)) + p64["gen_b2"].view(G, 1, 64)
  sig = torch.exp(p64["log_sigma"].view(G, 1, 64)) + 1e-3
  x = X64.repeat(S, 1).view(S * B, G, 64).permute(1, 0, 2)
  ll = (-0.5 * np.log(2 * np.pi) - torch.log(sig) - 0.5 * ((x - mu) / sig) ** 2).sum() / S / B
  theta = sum((-0.5 * np.log(2 * np.pi) - 0.5 * p64[n] ** 2).sum() for n in ("gen_W1", "gen_b1", "gen_W2", "gen_b2"))
  (kl - ll - theta).backward()

  err = {}
  for mode in ("fp32", "default"):
    if mode == "fp32":
      _hooks(OIVAE_ENC="fp32", OIVAE_DEC="v4")
    else:
      _hooks(OIVAE_ENC=None, OIVAE_DEC=None)
    eng._ws_key = None
    eng.forward_backward(X, eps, 0.7)
    torch.cuda.synchronize()
    e = {}
    for n in names:
      want = p64[n].grad.reshape(-1).cpu()
      got = eng.g[n].double().reshape(-1).cpu()
      assert torch.isfinite(got).all(), (mode, n)
      e[n] = float((got - want).abs().max() / want.abs().max().clamp(min=1e-30))   # x_zero: exact zeros expected
    err[mode] = e
  _hooks(OIVAE_ENC=None, OIVAE_DEC=None)
  # saturated_tanh: 1 - t^2 cancels in fp32 whoever computes it (the reference's tanh backward too); there the
  # fp32-instruction kernels' own error against float64 (3e-5) is the bar
  bar = 2e-5 if case != "saturated_tanh" else 5e-5
  for n in names:
    assert err["default"][n] <= bar, (case, n, err)
    assert err["default"][n] <= 2.0 * err["fp32"][n] + 1e-6, (case, n, err)
