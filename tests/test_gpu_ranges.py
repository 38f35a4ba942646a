"""The fp16-term kernels scale X by a power of two taken from max|X|.  Nobody can afford a pass over X to find it
(17 GB at the scale-up shape), so each kernel that streams X scales it by the maximum the previous pass left in the
WORKSPACE, records the maximum it reads itself, and a guard repeats the launch when the two powers of two differ
(csrc/x3_encoder.hip "ranges", mfma_decoder.hip range_guard).  Under test here: every step entry point is a function of
its own arguments -- no host-side memory of earlier calls (round 1 kept one), whatever the workspace holds, whichever
engine ran last on the device.  SURVEY 8(b): "no global state"; /root/reference/lib/oivae.py:42-90 is a pure function
of its inputs.  Tolerances: gradients within 5e-6 of their tensor's largest magnitude against float64 (the fp32 matrix
instructions reach 1e-6 .. 3e-6 on the same inputs, tests/test_gpu_x3.py)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G, K = 6, 64
DEC = ("gen_W1", "gen_b1", "gen_W2", "gen_b2", "log_sigma", "Ws")


def _engine(seed, groups=G):
  from tests.test_gpu_scale import _engine as scale_engine
  return scale_engine(groups, 64, K, 64, 64, seed=seed)


def _decoder_ref64(eng, X, z, B, S, groups=G):
  """float64 autograd of -loglik / (S B): lib/models.py:91-95, lib/distributions.py:40-45, lib/oivae.py:55-59"""
  N = B * S
  p64 = {n: eng.p[n].double().clone().requires_grad_(True) for n in DEC}
  z64 = z.double().clone().requires_grad_(True)
  h = torch.einsum("nk,gkh->gnh", z64, p64["Ws"].view(groups, K, 64))
  a1 = torch.einsum("gnh,gmh->gnm", h, p64["gen_W1"].view(groups, 64, 64)) + p64["gen_b1"].view(groups, 1, 64)
  mu = torch.einsum("gnm,gdm->gnd", torch.tanh(a1), p64["gen_W2"].view(groups, 64, 64)) + p64["gen_b2"].view(groups, 1, 64)
  sig = torch.exp(p64["log_sigma"].view(groups, 1, 64)) + 1e-3
  x = X.double().repeat(S, 1).view(N, groups, 64).permute(1, 0, 2)
  ll = (-0.5 * np.log(2 * np.pi) - torch.log(sig) - 0.5 * ((x - mu) / sig) ** 2).sum()
  (-ll / (S * B)).backward()
  want = {n: p64[n].grad.reshape(-1).cpu() for n in DEC}
  want["dz"] = z64.grad.cpu()
  return want


def _decoder(eng, X, z, B, S, ws):
  from oi_vae_amd import _lib
  eng.flat_grads.fill_(float("nan"))
  dz = torch.full((K, B * S), float("nan"), device="cuda")      # the matrix-core decoders hand dz over as [K, N]
  _lib.check(eng.lib.oivae_decoder_fwd_bwd_f32(ctypes.byref(eng._shape), ctypes.byref(eng._pt), ctypes.byref(eng._gt),
             X.data_ptr(), B, S, z.data_ptr(), 1e-3, dz.data_ptr(), ws.data_ptr(), eng._ws_bytes, eng._stream()), "decoder")
  torch.cuda.synchronize()
  got = {n: eng.g[n].double().reshape(-1).cpu() for n in DEC}
  got["dz"] = dz.t().double().cpu()
  return got


def _close(got, want, tol, what):
  for n in want:
    assert torch.isfinite(got[n]).all(), (what, n)
    e = float((got[n] - want[n]).abs().max() / want[n].abs().max())
    assert e <= tol, (what, n, e)


def _fwd(eng, X, eps, ws):
  from oi_vae_amd import _lib
  B, S = X.shape[0], eps.shape[0]
  mu, sig, ea = (torch.empty(B, K, device="cuda") for _ in range(3))
  z = torch.empty(S * B, K, device="cuda")
  pm, ps = torch.zeros(1, K, device="cuda"), torch.ones(1, K, device="cuda")
  _lib.check(eng.lib.oivae_encoder_fwd_f32(ctypes.byref(eng._shape), ctypes.byref(eng._pt), X.data_ptr(), B, S,
             eps.data_ptr(), 1.0, 1e-3, pm.data_ptr(), ps.data_ptr(), 1, mu.data_ptr(), sig.data_ptr(), ea.data_ptr(),
             z.data_ptr(), ws.data_ptr(), eng._ws_bytes, eng._stream()), "encoder_fwd")
  return z


@pytest.mark.parametrize("factor", [1000.0, 1.0 / 4096.0, 0.0])
def test_decoder_after_x_changed_in_place(factor):
  """Forward call on X, then X is overwritten IN PLACE (same pointer, same batch, same workspace) and the decoder is
  called without a new forward call: the ranges in the workspace are stale, the result must be the one of the new X."""
  B, S = 150, 2
  eng = _engine(41)
  g = torch.Generator(device="cuda").manual_seed(5)
  X = torch.rand(B, G * 64, generator=g, device="cuda")
  eps = torch.randn(S, B, K, generator=g, device="cuda")
  eng._ws_key = None
  ws = eng._workspace(B, S)
  z = _fwd(eng, X, eps, ws)
  z = torch.randn(B * S, K, generator=g, device="cuda")
  X.mul_(factor)
  if factor != 0.0:
    eng.p["log_sigma"].fill_(float(np.log(max(factor, 1.0))))   # keep (x - mu)/sigma in a sane range for the comparison
  want = _decoder_ref64(eng, X, z, B, S)
  _close(_decoder(eng, X, z, B, S, ws), want, 5e-6, factor)
  _close(_decoder(eng, X, z, B, S, ws), want, 5e-6, (factor, "second call: ranges corrected by the first"))


def test_decoder_bits_do_not_depend_on_a_stale_maximum_of_the_same_binade():
  """The guard accepts a recorded max|X| that shares its power of two with the maximum the kernel reads.  The scales
  must then depend on that power of two alone: a decoder call after a forward call of ANOTHER X with a different
  maximum in the same binade gives, bit for bit, what it gives after a forward call of its own X (every tensor but dz,
  whose float atomics are order-dependent in the last bits)."""
  B, S = 128, 1
  g = torch.Generator(device="cuda").manual_seed(8)
  X = torch.rand(B, G * 64, generator=g, device="cuda") * 0.6 + 0.05        # max just under 0.65: binade [0.5, 1)
  Xo = torch.rand(B, G * 64, generator=g, device="cuda") * 0.93 + 0.05     # max just under 0.98: same binade
  assert 0.5 <= float(X.max()) < float(Xo.max()) < 1.0
  eps = torch.randn(S, B, K, generator=g, device="cuda")
  z = torch.randn(B * S, K, generator=g, device="cuda")
  res = []
  for first in (X, Xo):
    eng = _engine(47)
    eng._ws_key = None
    ws = eng._workspace(B, S)
    ws.zero_()
    _fwd(eng, first, eps, ws)                    # leaves max|first| in the workspace's ranges
    res.append(_decoder(eng, X, z, B, S, ws))
  for n in DEC:
    assert torch.equal(res[0][n], res[1][n]), n
  _close(res[1], _decoder_ref64(eng, X, z, B, S), 5e-6, "stale same-binade maximum")


def test_decoder_on_a_workspace_that_never_saw_a_forward_call():
  B, S = 97, 1
  eng = _engine(43)
  g = torch.Generator(device="cuda").manual_seed(6)
  X = torch.rand(B, G * 64, generator=g, device="cuda") * 37.0
  z = torch.randn(B * S, K, generator=g, device="cuda")
  eng._ws_key = None
  ws = eng._workspace(B, S)
  want = _decoder_ref64(eng, X, z, B, S)
  for fill in (0x00, 0xFF, 0x7F, 0x42):          # zeros, NaN patterns, huge and ordinary floats as "ranges"
    ws.fill_(fill)
    _close(_decoder(eng, X, z, B, S, ws), want, 5e-6, hex(fill))


def test_encoder_backward_after_x_changed_in_place():
  """oivae_encoder_bwd_f32 with mu_q = 0, sig_q = 1, eps = 0 under a standard-normal prior: dWmu = dz^T X."""
  from oi_vae_amd import Engine, ModelSpec, _lib
  B, D = 4096, 8192
  eng = Engine(ModelSpec(D=D, K=K, H=64, M=64, group_dims=[64] * (D // 64), gen_type="mlp"), device="cuda:0")
  g = torch.Generator(device="cuda").manual_seed(8)
  for n in ("enc_Wmu", "enc_Wsig"):
    eng.p[n].copy_((torch.rand(eng.p[n].shape, generator=g, device="cuda") * 2 - 1) / D ** 0.5)
  X = torch.rand(B, D, generator=g, device="cuda")
  eps = torch.zeros(1, B, K, device="cuda")
  zeros, ones = torch.zeros(B, K, device="cuda"), torch.ones(B, K, device="cuda")
  pm, ps = torch.zeros(1, K, device="cuda"), torch.ones(1, K, device="cuda")
  dzT = torch.randn(K, B, generator=g, device="cuda") / B
  eng._ws_key = None
  ws = eng._workspace(B, 1)
  _fwd(eng, X, eps, ws)
  for factor in (1000.0, 1.0 / 65536.0, 1.0):
    X.mul_(factor)                                # in place: the workspace still holds the previous maximum
    want = (dzT.double() @ X.double()).cpu()
    eng.flat_grads.fill_(float("nan"))
    _lib.check(eng.lib.oivae_encoder_bwd_f32(ctypes.byref(eng._shape), ctypes.byref(eng._gt), X.data_ptr(), B, 1,
               eps.data_ptr(), 1.0, pm.data_ptr(), ps.data_ptr(), 1, zeros.data_ptr(), ones.data_ptr(), ones.data_ptr(),
               dzT.data_ptr(), ws.data_ptr(), eng._ws_bytes, eng._stream()), "encoder_bwd")
    torch.cuda.synchronize()
    got = eng.g["enc_Wmu"].double().cpu()
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) <= 5e-6 * float(want.abs().max()), factor


def test_two_engines_interleaved_on_one_device():
  """Two models of different size and data scale take turns on cuda:0, entry point by entry point; each must get what it
  gets alone (dz is a float-atomic sum: 1e-5 of scale, everything else follows from it)."""
  from oi_vae_amd import _lib
  engs = [_engine(51, groups=6), _engine(52, groups=5)]
  data = []
  for i, eng in enumerate(engs):
    g = torch.Generator(device="cuda").manual_seed(60 + i)
    B = 128 if i == 0 else 70
    X = torch.rand(B, eng.spec.D, generator=g, device="cuda") * (1.0 if i == 0 else 5000.0)
    if i == 1:
      eng.p["log_sigma"].fill_(float(np.log(5000.0)))
      for n in ("enc_Wmu", "enc_Wsig"):
        eng.p[n].mul_(1.0 / 5000.0)
    eps = torch.randn(1, B, K, generator=g, device="cuda")
    data.append((X, eps))
  alone = []
  for eng, (X, eps) in zip(engs, data):
    eng._ws_key = None
    eng.forward_backward(X, eps, 1.0)
    torch.cuda.synchronize()
    alone.append(eng.flat_grads.double().cpu().clone())
  # interleaved: both forward calls, then both decoder calls, then finalize + backward calls
  st = []
  for eng, (X, eps) in zip(engs, data):
    B = X.shape[0]
    ws = eng._workspace(B, 1)
    ea, mu, sig = (torch.empty(B, K, device="cuda") for _ in range(3))
    z = torch.empty(B, K, device="cuda")
    dz = torch.empty(B * K, device="cuda")
    pm, ps = torch.zeros(1, K, device="cuda"), torch.ones(1, K, device="cuda")
    st.append(dict(B=B, ws=ws, ea=ea, mu=mu, sig=sig, z=z, dz=dz, pm=pm, ps=ps))
    eng.flat_grads.fill_(float("nan"))
  for eng, (X, eps), s in zip(engs, data, st):
    _lib.check(eng.lib.oivae_encoder_fwd_f32(ctypes.byref(eng._shape), ctypes.byref(eng._pt), X.data_ptr(), s["B"], 1,
               eps.data_ptr(), 1.0, 1e-3, s["pm"].data_ptr(), s["ps"].data_ptr(), 1, s["mu"].data_ptr(), s["sig"].data_ptr(),
               s["ea"].data_ptr(), s["z"].data_ptr(), s["ws"].data_ptr(), eng._ws_bytes, eng._stream()), "fwd")
  for eng, (X, eps), s in reversed(list(zip(engs, data, st))):
    _lib.check(eng.lib.oivae_decoder_fwd_bwd_f32(ctypes.byref(eng._shape), ctypes.byref(eng._pt), ctypes.byref(eng._gt),
               X.data_ptr(), s["B"], 1, s["z"].data_ptr(), 1e-3, s["dz"].data_ptr(), s["ws"].data_ptr(), eng._ws_bytes,
               eng._stream()), "dec")
  for eng, (X, eps), s in zip(engs, data, st):
    _lib.check(eng.lib.oivae_finalize_f32(ctypes.byref(eng._shape), ctypes.byref(eng._pt), ctypes.byref(eng._gt), s["B"], 1,
               1.0, 1.0, eng.scalars.data_ptr(), s["ws"].data_ptr(), eng._ws_bytes, eng._stream()), "fin")
  for eng, (X, eps), s in reversed(list(zip(engs, data, st))):
    _lib.check(eng.lib.oivae_encoder_bwd_f32(ctypes.byref(eng._shape), ctypes.byref(eng._gt), X.data_ptr(), s["B"], 1,
               eps.data_ptr(), 1.0, s["pm"].data_ptr(), s["ps"].data_ptr(), 1, s["mu"].data_ptr(), s["sig"].data_ptr(),
               s["ea"].data_ptr(), s["dz"].data_ptr(), s["ws"].data_ptr(), eng._ws_bytes, eng._stream()), "bwd")
  torch.cuda.synchronize()
  for i, eng in enumerate(engs):
    got = eng.flat_grads.double().cpu()
    for name in eng.seg_off:
      o, n = eng.seg_off[name], eng.seg_numel[name]
      if n:
        a, b = got[o:o + n], alone[i][o:o + n]
        assert torch.isfinite(a).all(), (i, name)
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-30, (i, name)
