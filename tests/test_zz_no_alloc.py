"""oivae_encoder_bwd_noalloc_f32 (include/oivae_hip.h): the encoder backward entry point whose scratch lives in a static
device array instead of a caller workspace.  It runs the same kernels as oivae_encoder_bwd_f32, so its gradients must
equal that entry point's bit for bit -- on two calls in a row, the second with a smaller input (the array keeps the
first call's contents), at shapes that take each of the three kernel families, and it must refuse what does not fit."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ENC = ("enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig")


def _engine(D, K, group_dims, gen_type, H, M, seed):
  from oi_vae_amd import Engine, ModelSpec
  eng = Engine(ModelSpec(D=D, K=K, H=H, M=M, group_dims=group_dims, gen_type=gen_type), device="cuda:0")
  g = torch.Generator(device="cuda").manual_seed(seed)
  return eng, g


def _inputs(eng, B, S, g, prior_rows):
  K, D = eng.spec.K, eng.spec.D
  r = lambda *s: torch.randn(*s, generator=g, device="cuda")
  X = torch.rand(B, D, generator=g, device="cuda")
  eps = r(S, B, K)
  mu_q, sig_q, ea = r(B, K), torch.rand(B, K, generator=g, device="cuda") + 0.1, torch.rand(B, K, generator=g, device="cuda") + 0.05
  pm, ps = r(prior_rows, K) * 0.1, torch.rand(prior_rows, K, generator=g, device="cuda") + 0.5
  dz = r(S * B * K) / B                      # flat: read as [N, K] or [K, N] by both entry points alike
  return X, eps, mu_q, sig_q, ea, pm, ps, dz


def _both(eng, B, S, g, prior_rows=1):
  from oi_vae_amd import _lib
  X, eps, mu_q, sig_q, ea, pm, ps, dz = _inputs(eng, B, S, g, prior_rows)
  sh, gt = ctypes.byref(eng._shape), ctypes.byref(eng._gt)
  eng._ws_key = None
  ws = eng._workspace(B, S)
  eng.flat_grads.fill_(float("nan"))
  _lib.check(eng.lib.oivae_encoder_bwd_f32(sh, gt, X.data_ptr(), B, S, eps.data_ptr(), 0.7, pm.data_ptr(), ps.data_ptr(),
             prior_rows, mu_q.data_ptr(), sig_q.data_ptr(), ea.data_ptr(), dz.data_ptr(), ws.data_ptr(), eng._ws_bytes,
             eng._stream()), "oivae_encoder_bwd_f32")
  torch.cuda.synchronize()
  want = {n: eng.g[n].clone() for n in ENC}
  eng.flat_grads.fill_(float("nan"))
  free0 = torch.cuda.mem_get_info()[0]
  _lib.check(eng.lib.oivae_encoder_bwd_noalloc_f32(sh, gt, X.data_ptr(), B, S, eps.data_ptr(), 0.7, pm.data_ptr(),
             ps.data_ptr(), prior_rows, mu_q.data_ptr(), sig_q.data_ptr(), ea.data_ptr(), dz.data_ptr(), eng._stream()),
             "oivae_encoder_bwd_noalloc_f32")
  torch.cuda.synchronize()
  assert torch.cuda.mem_get_info()[0] == free0            # no device memory was taken
  for n in ENC:
    got = eng.g[n]
    assert torch.isfinite(got).all(), n
    assert torch.equal(got, want[n]), (n, float((got - want[n]).abs().max()))


@pytest.mark.parametrize("shape", ["generic_ragged", "fp32_matrix", "split_16bit"])
def test_equals_the_workspace_entry_point_twice_in_a_row(shape):
  if shape == "generic_ragged":      # mocap-like: ragged one-joint groups, K=8, ten samples, [B,K] prior
    dims = [3, 3, 3, 2, 1, 1, 2, 3, 1, 2, 1, 3]
    eng, g = _engine(sum(dims), 8, dims, "tanh_linear", 8, 0, seed=1)
    sizes = [(64, 10, 64), (37, 10, 1)]
  elif shape == "fp32_matrix":       # MEG-like: K=10, D a multiple of 4, fp32 matrix-core kernels
    dims = [186] * 12
    eng, g = _engine(sum(dims), 10, dims, "linear", 1, 0, seed=2)
    sizes = [(4096, 1, 1), (1000, 1, 1)]
  else:                              # 2K = 128, D % 128 == 0, B % 32 == 0: the 16-bit-term kernels and their image
    eng, g = _engine(8192, 64, [64] * 128, "mlp", 64, 64, seed=3)
    sizes = [(4096, 1, 1), (1024, 1, 1)]
  for B, S, prior_rows in sizes:     # the second call is smaller: stale scratch contents must not show
    _both(eng, B, S, g, prior_rows)


def test_refuses_what_does_not_fit_the_static_scratch():
  from oi_vae_amd import _lib
  eng, g = _engine(8192, 64, [64] * 128, "mlp", 64, 64, seed=4)
  B = 65536                          # dpre alone would need 32 MiB
  t = torch.zeros(1, device="cuda")
  p = t.data_ptr()
  rc = eng.lib.oivae_encoder_bwd_noalloc_f32(ctypes.byref(eng._shape), ctypes.byref(eng._gt), p, B, 1, p, 1.0, p, p, 1,
                                             p, p, p, p, eng._stream())
  assert rc == 4                     # OIVAE_E_WORKSPACE
  msg = eng.lib.oivae_last_error().decode()
  assert "static array" in msg and "oivae_encoder_bwd_f32" in msg
