"""ctypes loader for oracle/prox_oracle.c (TEST INFRASTRUCTURE ONLY).

Bit-exact CPU restatement of the reference's proximal step
(/root/reference/lib/models.py:106-111) including the MKL-VML square root the
reference's torch.sqrt resolves to; see the header of prox_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "prox_oracle.c")
_SO = os.path.join(_HERE, "_build", "libprox_oracle.so")
_TABLE = os.path.join(os.path.dirname(_HERE), "oi_vae_amd", "data", "rsqrt14_u16.bin")
_lib = None
_tab = None


def build(force=False):
  """Compile the C restatement with gcc (called by __graft_entry__.build())."""
  if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", _SRC, "-o", _SO, "-lm"])
  return _SO


def lib():
  global _lib, _tab
  if _lib is None:
    build()
    L = ctypes.CDLL(_SO)
    vp = ctypes.c_void_p
    L.oivae_oracle_mkl_sqrtf_array.argtypes = [vp, vp, ctypes.c_long, vp]
    L.oivae_oracle_rsqrt14_array.argtypes = [vp, vp, ctypes.c_long, vp]
    L.oivae_oracle_prox_f32.argtypes = [vp, ctypes.c_long, ctypes.c_int, ctypes.c_float, vp, vp, vp]
    L.oivae_oracle_prox_f32.restype = ctypes.c_int
    L.oivae_oracle_row_sumsq.argtypes = [vp, ctypes.c_int]
    L.oivae_oracle_row_sumsq.restype = ctypes.c_float
    L.oivae_oracle_prox_cols_f32.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, vp, vp, vp]
    L.oivae_oracle_prox_cols_f32.restype = ctypes.c_int
    _tab = np.fromfile(_TABLE, dtype=np.uint16)
    assert _tab.size == 65536
    _lib = L
  return _lib


def mkl_sqrt(x):
  L = lib()
  x = np.ascontiguousarray(x, np.float32)
  out = np.empty_like(x)
  L.oivae_oracle_mkl_sqrtf_array(x.ctypes.data, out.ctypes.data, x.size, _tab.ctypes.data)
  return out


def proximal_step(Ws, t):
  """Returns (new Ws [..,H] float32, zero mask [..] uint8, row norms [..])."""
  L = lib()
  W = np.ascontiguousarray(Ws, np.float32).copy()
  H = W.shape[-1]
  rows = W.size // H
  mask = np.zeros(W.shape[:-1], np.uint8)
  norms = np.zeros(W.shape[:-1], np.float32)
  rc = L.oivae_oracle_prox_f32(W.ctypes.data, rows, H, float(t), _tab.ctypes.data,
                               mask.ctypes.data, norms.ctypes.data)
  if rc != 0:
    raise ValueError("prox oracle: H=%d outside the pinned range" % H)
  return W, mask, norms


def proximal_step_cols(W, group_dims, t):
  """FirstLayerSparseDecoder.proximal_step (/root/reference/lib/models.py:65-71) on the
  maps stacked on rows, W [sum H_g, K].  Returns (new W, zero mask [G,K], column norms [G,K])."""
  L = lib()
  W = np.ascontiguousarray(W, np.float32).copy()
  K = W.shape[1]
  off = np.concatenate([[0], np.cumsum(group_dims)]).astype(np.int32)
  assert off[-1] == W.shape[0]
  G = len(group_dims)
  mask = np.zeros((G, K), np.uint8)
  norms = np.zeros((G, K), np.float32)
  rc = L.oivae_oracle_prox_cols_f32(W.ctypes.data, off.ctypes.data, G, K, float(t), _tab.ctypes.data,
                                    mask.ctypes.data, norms.ctypes.data)
  if rc != 0:
    raise ValueError("prox oracle: group shape outside the pinned range")
  return W, mask, norms
