"""ctypes binding of liboivae_hip.so (C ABI declared in include/oivae_hip.h).

The library is the product's compute path; there is no fallback.  If it is
missing, or there is no MI355X to run it on, every entry point raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OIVAE_LIB_PATH: another build of the same library (A/B timing of a kernel change on one box; tools/ only)
LIB_PATH = os.environ.get("OIVAE_LIB_PATH") or os.path.join(_HERE, "liboivae_hip.so")

c_f = ctypes.c_float
c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64
c_vp = ctypes.c_void_p

GEN_TYPES = {"linear": 0, "tanh_linear": 1, "mlp": 2}

# every symbol include/oivae_hip.h declares
SYMBOLS = [
  "oivae_init", "oivae_last_error", "oivae_version", "oivae_workspace_bytes",
  "oivae_prox_f32", "oivae_sgd_prox_f32", "oivae_prox_cols_f32", "oivae_encoder_fwd_f32",
  "oivae_decoder_fwd_bwd_f32", "oivae_encoder_bwd_f32", "oivae_finalize_f32",
  "oivae_adam_f32", "oivae_encoder_bwd_noalloc_f32", "oivae_dz_layout", "oivae_reload_test_hooks",
  "oivae_check_errors", "oivae_small_step_applies", "oivae_small_step_f32",
  "oivae_gamma_workspace_bytes", "oivae_gamma_logprob_f32",
]


class Shape(ctypes.Structure):
  _fields_ = [("D", c_i32), ("K", c_i32), ("G", c_i32), ("H", c_i32), ("M", c_i32),
              ("gen_type", c_i32), ("max_group_dim", c_i32), ("uniform_group_dim", c_i32),
              ("group_off", c_vp)]


class Tensors(ctypes.Structure):
  _fields_ = [(n, c_vp) for n in ("enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig",
                                  "gen_W1", "gen_b1", "gen_W2", "gen_b2", "log_sigma", "Ws")]


class OivaeError(RuntimeError):
  pass


_lib = None


def load():
  """Load the shared library (no GPU needed for loading).  Raises if absent."""
  global _lib
  if _lib is not None:
    return _lib
  _lib = bind(LIB_PATH)
  return _lib


def bind(path):
  """ctypes handle of a build of the library with every entry point's signature declared."""
  if not os.path.exists(path):
    raise OivaeError(
      "%s is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
      "or `make -C oi_vae_amd/csrc`. There is no CPU fallback." % path)
  L = ctypes.CDLL(path)
  PS, PT = ctypes.POINTER(Shape), ctypes.POINTER(Tensors)
  L.oivae_init.argtypes = [ctypes.c_int]
  L.oivae_init.restype = ctypes.c_int
  L.oivae_last_error.restype = ctypes.c_char_p
  L.oivae_version.restype = ctypes.c_char_p
  L.oivae_workspace_bytes.argtypes = [PS, c_i64, c_i32]
  L.oivae_workspace_bytes.restype = c_i64
  L.oivae_dz_layout.argtypes = [PS, c_i64, c_i32]
  L.oivae_dz_layout.restype = ctypes.c_int
  L.oivae_prox_f32.argtypes = [c_vp, c_i64, c_i32, c_f, c_vp, c_vp, c_vp]
  L.oivae_sgd_prox_f32.argtypes = [c_vp, c_vp, c_i64, c_i32, c_f, c_f, c_i32, c_vp, c_vp, c_vp]
  L.oivae_prox_cols_f32.argtypes = [c_vp, c_vp, c_i32, c_i32, c_i32, c_f, c_vp, c_vp, c_vp]
  L.oivae_encoder_fwd_f32.argtypes = [PS, PT, c_vp, c_i64, c_i32, c_vp, c_f, c_f, c_vp, c_vp, c_i32,
                                      c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]
  L.oivae_decoder_fwd_bwd_f32.argtypes = [PS, PT, PT, c_vp, c_i64, c_i32, c_vp, c_f, c_vp, c_vp, c_i64, c_vp]
  L.oivae_encoder_bwd_f32.argtypes = [PS, PT, c_vp, c_i64, c_i32, c_vp, c_f, c_vp, c_vp, c_i32,
                                      c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]
  L.oivae_encoder_bwd_noalloc_f32.argtypes = [PS, PT, c_vp, c_i64, c_i32, c_vp, c_f, c_vp, c_vp, c_i32,
                                              c_vp, c_vp, c_vp, c_vp, c_vp]
  L.oivae_finalize_f32.argtypes = [PS, PT, PT, c_i64, c_i32, c_f, c_f, c_vp, c_vp, c_i64, c_vp]
  L.oivae_adam_f32.argtypes = [c_vp, c_vp, c_vp, c_vp, c_i64, c_f, c_f, c_f, c_f, c_i32, c_f, c_vp]
  L.oivae_small_step_applies.argtypes = [PS, c_i64, c_i32]
  L.oivae_small_step_applies.restype = ctypes.c_int
  L.oivae_small_step_f32.argtypes = [PS, PT, PT, c_vp, c_i64, c_i32, c_vp, c_f, c_f, c_vp, c_vp, c_i32, c_f, c_f, c_f,
                                     c_vp, c_vp, c_vp, c_vp, c_i32, PT, PT, c_f, c_f, c_f, c_f, c_i32,
                                     c_f, c_f, c_i32, c_vp, c_vp, c_i32, c_vp, c_i64, c_vp]
  L.oivae_small_step_f32.restype = ctypes.c_int
  L.oivae_check_errors.argtypes = [c_vp]
  L.oivae_check_errors.restype = ctypes.c_int
  L.oivae_gamma_workspace_bytes.argtypes = []
  L.oivae_gamma_workspace_bytes.restype = c_i64
  L.oivae_gamma_logprob_f32.argtypes = [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]
  L.oivae_gamma_logprob_f32.restype = ctypes.c_int
  L.oivae_reload_test_hooks.argtypes = []
  L.oivae_reload_test_hooks.restype = None
  for name in ("oivae_prox_f32", "oivae_sgd_prox_f32", "oivae_prox_cols_f32", "oivae_encoder_fwd_f32", "oivae_decoder_fwd_bwd_f32",
               "oivae_encoder_bwd_f32", "oivae_finalize_f32", "oivae_adam_f32", "oivae_encoder_bwd_noalloc_f32"):
    getattr(L, name).restype = ctypes.c_int
  return L


def set_test_hooks(**env):
  """Set (value) or clear (None) the library's test-hook environment variables -- OIVAE_FORCE_GENERIC, OIVAE_ENC,
  OIVAE_DEC -- and make the loaded library read them again (it does not look at the environment per call)."""
  for k, v in env.items():
    if k not in ("OIVAE_FORCE_GENERIC", "OIVAE_ENC", "OIVAE_DEC"):
      raise KeyError(k)
    if v is None:
      os.environ.pop(k, None)
    else:
      os.environ[k] = str(v)
  load().oivae_reload_test_hooks()


def check(rc, what):
  if rc != 0:
    msg = load().oivae_last_error().decode("utf-8", "replace")
    raise OivaeError("%s failed (code %d): %s" % (what, rc, msg))


_inited = set()


def init_device(index):
  """oivae_init for a device, once."""
  L = load()
  if index not in _inited:
    check(L.oivae_init(int(index)), "oivae_init")
    _inited.add(index)
  return L
