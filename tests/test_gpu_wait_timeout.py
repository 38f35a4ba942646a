"""The role-split decoder kernels hand tiles between waves through LDS counters and wait for them with a bound
(csrc/dec_shared.h lds_wait_ge).  A wait that runs out must not pass silently (round 1: the wave went on and read an
unpublished tile, rc == 0): the wave raises the library's error word in host-mapped memory and the NEXT step entry point
on that device fails with OIVAE_E_HIP.  liboivae_hip_waittest.so is the library with the bound set to one poll
(-DOIVAE_TEST_WAIT_LIMIT=1, csrc/Makefile), so that every hand-over that is not already complete times out; the kernel
still drains (it reads whatever the tile holds, in bounds) and the host sees the error."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_a_timed_out_hand_over_fails_the_next_call():
  from oi_vae_amd import _lib
  from tests.test_gpu_scale import _engine
  path = os.path.join(os.path.dirname(_lib.LIB_PATH), "liboivae_hip_waittest.so")
  L = _lib.bind(path)
  assert L.oivae_init(0) == 0
  G, K, B = 4, 64, 4096
  eng = _engine(G, 64, K, 64, 64, seed=3)           # buffers and shapes; the calls below go to the test library
  g = torch.Generator(device="cuda").manual_seed(1)
  X = torch.rand(B, G * 64, generator=g, device="cuda")
  z = torch.randn(B, K, generator=g, device="cuda")
  dz = torch.empty(K, B, device="cuda")
  eng._ws_key = None
  ws = eng._workspace(B, 1)
  args = (ctypes.byref(eng._shape), ctypes.byref(eng._pt), ctypes.byref(eng._gt), X.data_ptr(), B, 1, z.data_ptr(),
          1e-3, dz.data_ptr(), ws.data_ptr(), eng._ws_bytes, eng._stream())
  assert L.oivae_decoder_fwd_bwd_f32(*args) == 0    # launches; its waits time out on the device
  torch.cuda.synchronize()
  rc = L.oivae_decoder_fwd_bwd_f32(*args)           # the next call on the device reports it
  assert rc == 3                                    # OIVAE_E_HIP
  assert b"bounded wait" in L.oivae_last_error()
  torch.cuda.synchronize()
  # the shipped library on the same inputs: no time-out, no error, before and after
  S = _lib.load()
  for _ in range(2):
    assert S.oivae_decoder_fwd_bwd_f32(*args) == 0, S.oivae_last_error()
    torch.cuda.synchronize()
  assert torch.isfinite(eng.flat_grads[eng.seg_off["gen_W1"]:]).all()


def test_update_kernels_queued_behind_a_failed_decoder_change_nothing():
  """The host runs ahead of the device: by the time a hand-over times out, Adam and the SGD + proximal step of the
  same step are usually queued already.  They read the device-resident error word and return without touching a
  parameter; the host hears of the failure at its next call (here: the explicit oivae_check_errors), after which the
  same update kernels work again."""
  from oi_vae_amd import _lib
  from tests.test_gpu_scale import _engine
  path = os.path.join(os.path.dirname(_lib.LIB_PATH), "liboivae_hip_waittest.so")
  L = _lib.bind(path)
  assert L.oivae_init(0) == 0
  G, K, B = 4, 64, 4096
  eng = _engine(G, 64, K, 64, 64, seed=5)
  g = torch.Generator(device="cuda").manual_seed(2)
  X = torch.rand(B, G * 64, generator=g, device="cuda")
  z = torch.randn(B, K, generator=g, device="cuda")
  dz = torch.empty(K, B, device="cuda")
  eng._ws_key = None
  ws = eng._workspace(B, 1)
  n = eng.adam_numel
  m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
  mask = torch.zeros(G * K, dtype=torch.uint8, device="cuda")
  eng.flat_grads.normal_(generator=g)
  torch.cuda.synchronize()
  before = eng.flat_params.clone()
  st = eng._stream()
  dec = (ctypes.byref(eng._shape), ctypes.byref(eng._pt), ctypes.byref(eng._gt), X.data_ptr(), B, 1, z.data_ptr(),
         1e-3, dz.data_ptr(), ws.data_ptr(), eng._ws_bytes, st)
  adam = (eng.flat_params.data_ptr(), eng.flat_grads.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-2, 0.9, 0.999, 1e-8, 1, 1.0, st)
  sgd = (eng.p["Ws"].data_ptr(), eng.g["Ws"].data_ptr(), G * K, 64, 1e-2, 0.5, 1, mask.data_ptr(), None, st)
  # no synchronisation between the three calls: the update kernels are queued behind the failing launch
  assert L.oivae_decoder_fwd_bwd_f32(*dec) == 0
  rc_a, rc_s = L.oivae_adam_f32(*adam), L.oivae_sgd_prox_f32(*sgd)
  assert L.oivae_check_errors(st) == 3 or 3 in (rc_a, rc_s)   # reported at the first look after the kernel ran
  assert b"bounded wait" in L.oivae_last_error()
  torch.cuda.synchronize()
  assert torch.equal(eng.flat_params, before), "parameters changed behind a failed decoder launch"
  assert float(m.abs().sum()) == 0.0 and float(v.abs().sum()) == 0.0
  # words cleared: the same update kernels run now
  assert L.oivae_check_errors(st) == 0
  assert L.oivae_adam_f32(*adam) == 0 and L.oivae_sgd_prox_f32(*sgd) == 0
  torch.cuda.synchronize()
  assert not torch.equal(eng.flat_params[:n], before[:n]) and not torch.equal(eng.p["Ws"].reshape(-1), before[n:n + eng.p["Ws"].numel()])
