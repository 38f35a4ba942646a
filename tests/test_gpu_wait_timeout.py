"""The role-split decoder kernels hand tiles between waves through LDS counters and wait for them with a bound
(csrc/mfma_decoder.hip lds_wait_ge).  A wait that runs out must not pass silently (round 1: the wave went on and read an
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
