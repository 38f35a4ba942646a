"""Device-resident state and kernel sequencing of one oi-VAE on one MI355X.

The engine owns two flat fp32 device buffers, parameters and gradients, laid
out in the order the data-parallel gradient bucket uses
(SURVEY.md section 8e):

    [enc_Wmu | enc_bmu | enc_Wsig | enc_bsig | gen_W1 | gen_b1 | gen_W2 | gen_b2 | log_sigma | Ws]

Every tensor the reference's API exposes (nn.Linear weights of the encoder and
of each group generator, each group's log_sigma, Ws) is a *view* into these
buffers, so caller-owned torch optimizers update the engine's storage in place
and one all-reduce over ``flat_grads`` covers every gradient.

The step itself is the kernel sequence of include/oivae_hip.h; torch is used
for memory, streams and collectives only.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib

_SEG_ORDER = ("enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig", "gen_W1", "gen_b1", "gen_W2", "gen_b2",
              "log_sigma", "Ws")
_ALIGN = 64  # floats (256 B): keeps every segment 16-byte aligned for vector access


class ModelSpec:
  """Shapes of one oi-VAE (names follow SURVEY.md section 8)."""

  def __init__(self, D, K, H, group_dims, gen_type, M=0):
    self.D, self.K, self.H, self.M = int(D), int(K), int(H), int(M)
    self.group_dims = [int(d) for d in group_dims]
    self.G = len(self.group_dims)
    self.gen_type = gen_type
    if gen_type not in _lib.GEN_TYPES:
      raise ValueError("unknown generator type %r" % (gen_type,))
    if sum(self.group_dims) != self.D:
      raise ValueError("group dims sum to %d, not D=%d" % (sum(self.group_dims), self.D))
    if (gen_type == "mlp") != (self.M > 0):
      raise ValueError("M must be > 0 exactly for gen_type='mlp'")
    self.Hin = self.M if gen_type == "mlp" else self.H

  def seg_shapes(self) -> Dict[str, tuple]:
    s = self
    return {
      "enc_Wmu": (s.K, s.D), "enc_bmu": (s.K,), "enc_Wsig": (s.K, s.D), "enc_bsig": (s.K,),
      "gen_W1": (s.G, s.M, s.H) if s.gen_type == "mlp" else (0,),
      "gen_b1": (s.G, s.M) if s.gen_type == "mlp" else (0,),
      "gen_W2": (s.D, s.Hin), "gen_b2": (s.D,), "log_sigma": (s.D,), "Ws": (s.G, s.K, s.H),
    }

  def to_dict(self):
    return dict(D=self.D, K=self.K, H=self.H, M=self.M, group_dims=list(self.group_dims), gen_type=self.gen_type)


def require_gpu(device=None) -> torch.device:
  if not torch.cuda.is_available():
    raise _lib.OivaeError(
      "oi_vae_amd runs its training step on an MI355X through liboivae_hip.so; "
      "no GPU is visible and there is no CPU fallback.")
  dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
  if dev.type != "cuda":
    raise _lib.OivaeError("oi_vae_amd needs a cuda (ROCm) device, got %s" % dev)
  if dev.index is None:
    dev = torch.device("cuda", torch.cuda.current_device())
  return dev


class Engine:
  def __init__(self, spec: ModelSpec, device=None):
    self.spec = spec
    self.device = require_gpu(device)
    self.lib = _lib.init_device(self.device.index)
    shapes = spec.seg_shapes()
    self.seg_off: Dict[str, int] = {}
    self.seg_numel: Dict[str, int] = {}
    off = 0
    for name in _SEG_ORDER:
      n = int(np.prod(shapes[name]))
      self.seg_off[name] = off
      self.seg_numel[name] = n
      off += (n + _ALIGN - 1) // _ALIGN * _ALIGN
    self.flat_numel = off
    self.adam_numel = self.seg_off["Ws"]          # everything before Ws is Adam's
    self.flat_params = torch.zeros(off, dtype=torch.float32, device=self.device)
    self.flat_grads = torch.zeros(off, dtype=torch.float32, device=self.device)
    self.p = {n: self._view(self.flat_params, n, shapes[n]) for n in _SEG_ORDER}
    self.g = {n: self._view(self.flat_grads, n, shapes[n]) for n in _SEG_ORDER}
    offs = np.concatenate([[0], np.cumsum(spec.group_dims)]).astype(np.int32)
    self.group_off_host = offs
    self.group_off = torch.from_numpy(offs).to(self.device)
    self._shape = _lib.Shape(
      D=spec.D, K=spec.K, G=spec.G, H=spec.H, M=spec.M, gen_type=_lib.GEN_TYPES[spec.gen_type],
      max_group_dim=max(spec.group_dims),
      uniform_group_dim=spec.group_dims[0] if len(set(spec.group_dims)) == 1 else 0,
      group_off=self.group_off.data_ptr())
    self._pt = self._tensors(self.p)
    self._gt = self._tensors(self.g)
    self._ws: Optional[torch.Tensor] = None
    self._ws_key = None
    self.scalars = torch.zeros(8, dtype=torch.float32, device=self.device)
    self.zero_mask = torch.zeros(spec.G, spec.K, dtype=torch.uint8, device=self.device)
    self.row_norms = torch.zeros(spec.G, spec.K, dtype=torch.float32, device=self.device)
    self._bufs: Dict[str, torch.Tensor] = {}
    # Adam state for the fused optimizer path
    self.adam_m: Optional[torch.Tensor] = None
    self.adam_v: Optional[torch.Tensor] = None
    self.adam_t = 0
    # optional per-entry-point timing: a list that receives (name, start_event, end_event)
    self.trace = None
    self.trace_names = None   # with trace set: the entry points that get events (None = all of them)
    self.gen_sig_off = 1e-3
    self.bound = []  # (user tensor, grad view) pairs set by adopt()
    self._eval_grads = None
    self._eval_gt = None
    self._adam_mt = None   # (adam_m it was built for, Tensors of the moments' views) for small_step

  def _timed(self, name, fn):
    if self.trace is None or (self.trace_names is not None and name not in self.trace_names):
      return fn()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record(torch.cuda.current_stream(self.device))
    rc = fn()
    b.record(torch.cuda.current_stream(self.device))
    self.trace.append((name, a, b))
    return rc

  # ---- storage ---------------------------------------------------------------
  def _view(self, flat, name, shape):
    n = self.seg_numel[name]
    o = self.seg_off[name]
    return flat[o:o + n].view(*shape) if n else flat[o:o]

  @staticmethod
  def _tensors(d) -> _lib.Tensors:
    return _lib.Tensors(**{n: (d[n].data_ptr() if d[n].numel() else None) for n in _SEG_ORDER})

  def group_slices(self, name: str, g: int, grads=False) -> torch.Tensor:
    """View of group g's part of a generator tensor, shaped as the reference's
    per-group nn.Linear / log_sigma tensors."""
    src = self.g if grads else self.p
    s = self.spec
    a, b = int(self.group_off_host[g]), int(self.group_off_host[g + 1])
    if name in ("gen_W2",):
      return src[name][a:b]
    if name in ("gen_b2", "log_sigma"):
      return src[name][a:b]
    if name in ("gen_W1", "gen_b1", "Ws"):
      return src[name][g]
    raise KeyError(name)

  def load_params(self, params: Dict[str, object]):
    """params in the oracle's layout (lists per group for generator tensors)."""
    self._import(self.p, params, with_Ws=True)

  def _import(self, dst, params, with_Ws):
    s = self.spec

    def put(d, arr):
      t = torch.as_tensor(np.asarray(arr, np.float32)) if not torch.is_tensor(arr) else arr.detach().float()
      d.copy_(t.reshape(d.shape))

    for n in ("enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig") + (("Ws",) if with_Ws else ()):
      put(dst[n], params[n])
    for n in ("gen_W2", "gen_b2", "log_sigma"):
      put(dst[n], np.concatenate([np.asarray(a, np.float32) for a in params[n]], axis=0))
    if s.gen_type == "mlp":
      put(dst["gen_W1"], np.stack([np.asarray(a, np.float32) for a in params["gen_W1"]]))
      put(dst["gen_b1"], np.stack([np.asarray(a, np.float32) for a in params["gen_b1"]]))

  def load_adam_state(self, exp_avg: Dict[str, object], exp_avg_sq: Dict[str, object], steps_done: int):
    """The fused Adam's moments and step count from torch.optim.Adam's per-parameter state (the oracle's layout):
    resumes a run -- or runs one step teacher-forced -- from a state recorded elsewhere."""
    n = self.adam_numel
    self.adam_m = torch.zeros(n, dtype=torch.float32, device=self.device)
    self.adam_v = torch.zeros(n, dtype=torch.float32, device=self.device)
    shapes = self.spec.seg_shapes()
    for flat, src in ((self.adam_m, exp_avg), (self.adam_v, exp_avg_sq)):
      views = {k: self._view(flat, k, shapes[k]) for k in _SEG_ORDER if k != "Ws"}
      self._import(views, src, with_Ws=False)
    self.adam_t = int(steps_done)

  def _export(self, src) -> Dict[str, object]:
    s = self.spec
    out: Dict[str, object] = {n: src[n].detach().cpu().numpy().copy() for n in ("enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig", "Ws")}
    offs = self.group_off_host
    for n in ("gen_W2", "gen_b2", "log_sigma"):
      full = src[n].detach().cpu().numpy()
      out[n] = [full[offs[g]:offs[g + 1]].copy() for g in range(s.G)]
    if s.gen_type == "mlp":
      out["gen_W1"] = [a.copy() for a in src["gen_W1"].detach().cpu().numpy()]
      out["gen_b1"] = [a.copy() for a in src["gen_b1"].detach().cpu().numpy()]
    return out

  def export_params(self):
    return self._export(self.p)

  def export_grads(self):
    return self._export(self.g)

  # ---- helpers ---------------------------------------------------------------
  def _stream(self):
    return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

  def _workspace(self, B, S):
    self._small_ticket = None   # whoever asks for the workspace may write it: only small_step re-arms this (see there)
    key = (B, S)
    if self._ws_key != key:
      nbytes = self.lib.oivae_workspace_bytes(ctypes.byref(self._shape), B, S)
      if nbytes < 0:
        raise _lib.OivaeError("oivae_workspace_bytes: %s" % self.lib.oivae_last_error().decode())
      if self._ws is None or self._ws.numel() < nbytes:
        self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
      self._ws_key = key
      self._ws_bytes = int(nbytes)
    return self._ws

  def _buf(self, name, shape):
    t = self._bufs.get(name)
    n = int(np.prod(shape))
    if t is None or t.numel() < n:
      t = torch.empty(n, dtype=torch.float32, device=self.device)
      self._bufs[name] = t
    return t[:n].view(*shape)

  def _dev_f32(self, x, what):
    if not torch.is_tensor(x):
      x = torch.as_tensor(np.asarray(x, np.float32))
    if x.dtype != torch.float32:
      x = x.float()
    if x.device != self.device:
      x = x.to(self.device, non_blocking=True)
    return x.contiguous()

  # ---- the step ----------------------------------------------------------------
  def forward_backward(self, X, eps, c, sig_off=1e-3, gen_sig_off=1e-3, prior_mu=None, prior_sigma=None,
                       sigma_theta=1.0, lam=1.0, grad_sync=None, evaluate_only=False):
    """Forward + backward of loss = z_kl - loglik_term - logprob_theta
    (/root/reference/lib/oivae.py:43-75).  Fills ``flat_grads`` completely and
    returns mu_q, sig_q, z (fresh tensors) plus the 8-float scalar block.

    Kernel order: encoder forward, grouped decoder forward+backward, weight prior +
    penalty + scalars (they need nothing from the encoder backward), encoder
    backward.  ``grad_sync`` (parallel.BucketedGradSync) is told when the decoder's
    and then the encoder's gradient segments are final, so that the data-parallel
    all-reduce of the first overlaps the encoder backward.

    ``evaluate_only``: held-out evaluation (the log-likelihood the reference leaves
    stubbed at meg_oi_vae.py:258-260).  The same kernels run, but every gradient goes
    to a scratch image of ``flat_grads`` and the encoder backward is skipped, so
    parameters, ``flat_grads`` and optimizer state are untouched."""
    s = self.spec
    with torch.cuda.device(self.device):
      X = self._dev_f32(X, "X")
      if X.dim() != 2 or X.size(1) != s.D:
        raise ValueError("X must be [B, %d], got %s" % (s.D, tuple(X.shape)))
      B = X.size(0)
      eps = self._dev_f32(eps, "eps")
      if eps.dim() != 3 or eps.size(1) != B or eps.size(2) != s.K:
        raise ValueError("eps must be [S, %d, %d], got %s" % (B, s.K, tuple(eps.shape)))
      S = eps.size(0)
      if prior_mu is None:
        prior_mu = self._const("prior0", 0.0)
        prior_sigma = self._const("prior1", 1.0)
      prior_mu = self._dev_f32(prior_mu, "prior_mu").view(-1, s.K)
      prior_sigma = self._dev_f32(prior_sigma, "prior_sigma").view(-1, s.K)
      prior_rows = prior_mu.size(0)
      if prior_rows not in (1, B) or prior_sigma.size(0) != prior_rows:
        raise ValueError("prior_z must have 1 or B=%d rows, got %d" % (B, prior_rows))
      N = B * S
      ws = self._workspace(B, S)
      mu_q = torch.empty(B, s.K, dtype=torch.float32, device=self.device)
      sig_q = torch.empty(B, s.K, dtype=torch.float32, device=self.device)
      z = torch.empty(N, s.K, dtype=torch.float32, device=self.device)
      ea = self._buf("ea", (B, s.K))
      dz = self._buf("dz", (N, s.K))
      st = self._stream()
      L = self.lib
      sh, pt, gt = ctypes.byref(self._shape), ctypes.byref(self._pt), ctypes.byref(self._gt)
      if evaluate_only:
        if self._eval_gt is None:
          self._eval_grads = torch.zeros_like(self.flat_grads)
          shapes = s.seg_shapes()
          self._eval_gt = self._tensors({n: self._view(self._eval_grads, n, shapes[n]) for n in _SEG_ORDER})
        gt = ctypes.byref(self._eval_gt)
      _lib.check(self._timed("encoder_fwd", lambda: L.oivae_encoder_fwd_f32(
        sh, pt, X.data_ptr(), B, S, eps.data_ptr(), float(c), float(sig_off),
        prior_mu.data_ptr(), prior_sigma.data_ptr(), prior_rows,
        mu_q.data_ptr(), sig_q.data_ptr(), ea.data_ptr(), z.data_ptr(),
        ws.data_ptr(), self._ws_bytes, st)), "oivae_encoder_fwd_f32")
      _lib.check(self._timed("decoder", lambda: L.oivae_decoder_fwd_bwd_f32(
        sh, pt, gt, X.data_ptr(), B, S, z.data_ptr(), float(gen_sig_off),
        dz.data_ptr(), ws.data_ptr(), self._ws_bytes, st)), "oivae_decoder_fwd_bwd_f32")
      def finalize(stream_ptr):
        _lib.check(self._timed("finalize", lambda: L.oivae_finalize_f32(
          sh, pt, gt, B, S, float(sigma_theta), float(lam), self.scalars.data_ptr(),
          ws.data_ptr(), self._ws_bytes, stream_ptr)), "oivae_finalize_f32")
      if evaluate_only:
        finalize(st)
        return dict(mu_q=mu_q, sig_q=sig_q, z=z, scalars=self.scalars, X=X, eps=eps)
      # (Tried: these three small kernels on a side stream BESIDE the encoder backward's own small kernels -- dpre, bias
      # sums, column maxima, the image of dpre^T -- so that the two idle stretches overlap: 2.545 against 2.513 ms per step
      # at 8192 rows, slower; the two cross-stream event waits cost more than the 30 us they hide.  A HIP graph of the
      # whole sequence replays in exactly the eager time, tools/graph_probe.py: the host is never the bound.)
      finalize(st)
      if grad_sync is not None:
        grad_sync.decoder_grads_ready()
      _lib.check(self._timed("encoder_bwd", lambda: L.oivae_encoder_bwd_f32(
        sh, gt, X.data_ptr(), B, S, eps.data_ptr(), float(c),
        prior_mu.data_ptr(), prior_sigma.data_ptr(), prior_rows,
        mu_q.data_ptr(), sig_q.data_ptr(), ea.data_ptr(), dz.data_ptr(),
        ws.data_ptr(), self._ws_bytes, st)), "oivae_encoder_bwd_f32")
      if grad_sync is not None:
        grad_sync.encoder_grads_ready()
    return dict(mu_q=mu_q, sig_q=sig_q, z=z, scalars=self.scalars, X=X, eps=eps)

  def check_errors(self):
    """Waits for everything queued on the current stream and raises OivaeError if a kernel reported a failure on the
    device (a decoder hand-over that timed out).  The step itself never waits: parameters are protected on the
    device (the update kernels refuse to run behind a failed launch) and the host hears of it at its next call;
    this is the definite answer, e.g. behind the last step of a run."""
    with torch.cuda.device(self.device):
      _lib.check(self.lib.oivae_check_errors(self._stream()), "oivae_check_errors")

  def small_step_applies(self, B, S) -> bool:
    """Whether the one-launch step (oivae_small_step_f32: the scripts' own small shapes) takes this model and batch."""
    return bool(self.lib.oivae_small_step_applies(ctypes.byref(self._shape), int(B), int(S)))

  def small_step(self, X, eps, c, prox_step_size, lr, Ws_lr, betas=(0.9, 0.999), adam_eps=1e-8, do_prox=True,
                 optimizers=True, sig_off=1e-3, gen_sig_off=1e-3, prior_mu=None, prior_sigma=None, sigma_theta=1.0, lam=1.0):
    """The whole step in one launch (include/oivae_hip.h oivae_small_step_f32): forward, backward, scalars and -- with
    `optimizers` -- Adam, SGD on Ws and the proximal step.  Returns what forward_backward returns (pre-update values)."""
    s = self.spec
    with torch.cuda.device(self.device):
      X = self._dev_f32(X, "X")
      eps = self._dev_f32(eps, "eps")
      B, S = X.size(0), eps.size(0)
      if X.dim() != 2 or X.size(1) != s.D or eps.dim() != 3 or eps.size(1) != B or eps.size(2) != s.K:
        raise ValueError("X must be [B, %d] and eps [S, B, %d]" % (s.D, s.K))
      if prior_mu is None:
        prior_mu, prior_sigma = self._const("prior0", 0.0), self._const("prior1", 1.0)
      prior_mu = self._dev_f32(prior_mu, "prior_mu").view(-1, s.K)
      prior_sigma = self._dev_f32(prior_sigma, "prior_sigma").view(-1, s.K)
      prior_rows = prior_mu.size(0)
      if prior_rows not in (1, B) or prior_sigma.size(0) != prior_rows:
        raise ValueError("prior_z must have 1 or B=%d rows, got %d" % (B, prior_rows))
      last_ticket = getattr(self, "_small_ticket", None)
      ws = self._workspace(B, S)
      mu_q = torch.empty(B, s.K, dtype=torch.float32, device=self.device)
      sig_q = torch.empty(B, s.K, dtype=torch.float32, device=self.device)
      z = torch.empty(B * S, s.K, dtype=torch.float32, device=self.device)
      if optimizers:
        if self.adam_m is None:
          self.adam_m = torch.zeros(self.adam_numel, dtype=torch.float32, device=self.device)
          self.adam_v = torch.zeros(self.adam_numel, dtype=torch.float32, device=self.device)
        if self._adam_mt is None or self._adam_mt[0] is not self.adam_m:
          shapes = s.seg_shapes()
          def tens(flat):
            return self._tensors({n: (self._view(flat, n, shapes[n]) if n != "Ws" else flat[0:0]) for n in _SEG_ORDER})
          self._adam_mt = (self.adam_m, tens(self.adam_m), tens(self.adam_v))
        self.adam_t += 1
        am, av = ctypes.byref(self._adam_mt[1]), ctypes.byref(self._adam_mt[2])
      else:
        am = av = None
      # the kernel's workgroup ticket lives in the workspace; the library zeroes it first (a memset node) unless told that
      # this very workspace last went through a completed small step of the same B and S (include/oivae_hip.h)
      # (every other user of the workspace goes through _workspace(), which disarms it)
      key = (ws.data_ptr(), B, S)
      ticket_ok = 1 if last_ticket == key else 0
      _lib.check(self._timed("small_step", lambda: self.lib.oivae_small_step_f32(
        ctypes.byref(self._shape), ctypes.byref(self._pt), ctypes.byref(self._gt), X.data_ptr(), B, S, eps.data_ptr(),
        float(c), float(sig_off), prior_mu.data_ptr(), prior_sigma.data_ptr(), prior_rows, float(gen_sig_off),
        float(sigma_theta), float(lam), mu_q.data_ptr(), sig_q.data_ptr(), z.data_ptr(), self.scalars.data_ptr(),
        1 if optimizers else 0, am, av, float(lr), float(betas[0]), float(betas[1]), float(adam_eps), max(self.adam_t, 1),
        float(Ws_lr), float(prox_step_size), 1 if do_prox else 0, self.zero_mask.data_ptr(), self.row_norms.data_ptr(),
        ticket_ok, ws.data_ptr(), self._ws_bytes, self._stream())), "oivae_small_step_f32")
      self._small_ticket = key     # the call was accepted and queued: its kernel leaves the ticket at zero
    return dict(mu_q=mu_q, sig_q=sig_q, z=z, scalars=self.scalars, X=X, eps=eps)

  def _const(self, name, val):
    t = self._bufs.get(name)
    if t is None:
      t = torch.full((1, self.spec.K), val, dtype=torch.float32, device=self.device)
      self._bufs[name] = t
    return t

  def proximal_step(self, t: float):
    """lib/models.py:106-111 on the engine's Ws, in place."""
    s = self.spec
    with torch.cuda.device(self.device):
      _lib.check(self.lib.oivae_prox_f32(self.p["Ws"].data_ptr(), s.G * s.K, s.H, float(t),
                                         self.zero_mask.data_ptr(), self.row_norms.data_ptr(), self._stream()),
                 "oivae_prox_f32")

  def sgd_prox_step(self, lr: float, t: float, do_prox: bool = True):
    """Ws -= lr*dWs then the proximal step, one kernel."""
    s = self.spec
    with torch.cuda.device(self.device):
      _lib.check(self.lib.oivae_sgd_prox_f32(self.p["Ws"].data_ptr(), self.g["Ws"].data_ptr(), s.G * s.K, s.H,
                                             float(lr), float(t), 1 if do_prox else 0,
                                             self.zero_mask.data_ptr(), self.row_norms.data_ptr(), self._stream()),
                 "oivae_sgd_prox_f32")

  def adam_step(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, grad_scale: float = 1.0, part=None):
    """torch.optim.Adam (default options) over every non-Ws parameter, one kernel.  ``part`` splits the update along
    the data-parallel buckets (Adam is elementwise: the two halves give the bits of the whole): "dec" -- generators
    and log_sigma, FIRST, it opens the step -- then "enc" -- the encoder, once its gradients are reduced."""
    n = self.adam_numel
    if self.adam_m is None:
      self.adam_m = torch.zeros(n, dtype=torch.float32, device=self.device)
      self.adam_v = torch.zeros(n, dtype=torch.float32, device=self.device)
    cut = self.seg_off["gen_W1"]
    if part not in (None, "dec", "enc"):
      raise ValueError("part must be None, 'dec' or 'enc'")
    if part != "enc":
      self.adam_t += 1
    lo, hi = (0, n) if part is None else ((cut, n) if part == "dec" else (0, cut))
    if hi <= lo:
      return
    with torch.cuda.device(self.device):
      _lib.check(self.lib.oivae_adam_f32(self.flat_params.data_ptr() + 4 * lo, self.flat_grads.data_ptr() + 4 * lo,
                                         self.adam_m.data_ptr() + 4 * lo, self.adam_v.data_ptr() + 4 * lo, hi - lo, float(lr),
                                         float(betas[0]), float(betas[1]), float(eps), self.adam_t,
                                         float(grad_scale), self._stream()), "oivae_adam_f32")
