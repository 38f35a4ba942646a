"""Engine-native optimizers with torch.optim's constructor shape.

`Adam` and `SGD` take the same param-group lists the reference's scripts pass to
torch.optim (bars_data_oivae.py:233-244).  Once the parameters live in an
engine (after `OIVAE(...)` adopted the model) they update them with ONE fused
HIP kernel each over the engine's flat buffers instead of one launch per
tensor; `SGD` on `Ws` is fused with the proximal step that follows it.
torch.optim optimizers keep working too (the parameters are ordinary views).
"""
import torch


class _EngineOptimizer(object):
  def __init__(self, params, **defaults):
    groups = list(params)
    if groups and not isinstance(groups[0], dict):
      groups = [{"params": groups}]
    self.param_groups = []
    for g in groups:
      extra = sorted(k for k in g if k != "params" and k not in defaults)
      if extra:
        # one fused kernel with one set of options: anything it would silently ignore is refused
        raise ValueError("%s: unsupported per-group option(s) %s (supported: %s); use torch.optim for them"
                         % (type(self).__name__, ", ".join(extra), ", ".join(sorted(defaults))))
      d = dict(defaults)
      d.update({k: v for k, v in g.items() if k != "params"})
      d["params"] = list(g["params"])
      self.param_groups.append(d)
    self._engine = None

  def _uniform(self, key):
    """The one value of an option across all param groups (read at step time: schedulers may change it)."""
    vals = {(tuple(g[key]) if isinstance(g[key], (list, tuple)) else g[key]) for g in self.param_groups}
    if len(vals) != 1:
      raise ValueError("engine-native optimizers apply one %r to every param group, got %s" % (key, sorted(vals)))
    return vals.pop()

  def zero_grad(self, set_to_none=True):
    """Attached to an engine: nothing to do -- every parameter's .grad is a view of the engine's flat gradient buffer,
    bound once, and each step overwrites that buffer completely (no loop over thousands of per-group tensors)."""
    if self._engine is not None:
      return
    for g in self.param_groups:
      for p in g["params"]:
        p.grad = None

  def _attach(self, engine):
    self._engine = engine

  def _lr(self):
    return self._uniform("lr")


class Adam(_EngineOptimizer):
  """torch.optim.Adam with default options over every non-Ws parameter."""

  def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
    if weight_decay != 0 or amsgrad:
      raise ValueError("oi_vae_amd.optim.Adam is torch.optim.Adam with default options (weight_decay=0, "
                       "amsgrad=False): the fused kernel implements nothing else")
    super().__init__(params, lr=lr, betas=betas, eps=eps)

  def covers(self, engine):
    ptrs = {p.data_ptr() for g in self.param_groups for p in g["params"]}
    want = {t.data_ptr() for t, _ in engine.bound if t.data_ptr() != engine.p["Ws"].data_ptr()}
    return ptrs == want

  def step(self):
    if self._engine is None:
      raise RuntimeError("oi_vae_amd.optim.Adam steps engine parameters: pass it to OIVAE(optimizers=[...])")
    self._engine.adam_step(self._lr(), betas=self._uniform("betas"), eps=self._uniform("eps"))


class SGD(_EngineOptimizer):
  """Plain SGD (momentum 0) on Ws; OIVAE fuses it with the proximal step."""

  def __init__(self, params, lr=1e-3, momentum=0):
    if momentum != 0:
      raise ValueError("the reference uses momentum=0 on Ws (bars_data_oivae.py:243); nothing else is fused")
    super().__init__(params, lr=lr, momentum=0)
    self.fused_with_prox = False

  def covers(self, engine):
    ptrs = [p.data_ptr() for g in self.param_groups for p in g["params"]]
    return ptrs == [engine.p["Ws"].data_ptr()]

  def step(self):
    if self._engine is None:
      raise RuntimeError("oi_vae_amd.optim.SGD steps engine parameters: pass it to OIVAE(optimizers=[...])")
    if self.fused_with_prox:
      return  # OIVAE.step runs oivae_sgd_prox_f32 right after the optimizers
    self._engine.sgd_prox_step(self._lr(), 0.0, do_prox=False)
