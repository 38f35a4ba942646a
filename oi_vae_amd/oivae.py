"""`OIVAE` and `NormalPriorTheta` with the reference's interface
(/root/reference/lib/oivae.py:8-90); the step runs on the MI355X engine."""
import torch

from . import _lib, parallel
from .adopt import adopt, probe_encoder_head
from .distributions import Normal
from .optim import SGD as _EngineSGD, _EngineOptimizer


class NormalPriorTheta(object):
  """Zero-mean Normal prior on every parameter of every group generator's
  mu_net (lib/oivae.py:8-23).  Inside `OIVAE.step` the value and its gradient
  come from `oivae_finalize_f32`; `logprob` is kept for callers outside it."""

  def __init__(self, sigma):
    self.sigma = sigma

  def logprob(self, module):
    """Sum over every parameter tensor of every group generator of log N(theta; 0, sigma): one
    closed-form term per tensor, -n (0.5 LOG2PI + log sigma) - 0.5 sum(theta^2) / sigma^2."""
    from .distributions import LOG2PI
    total = 0
    for gen in module.group_generators:
      for theta in gen.parameters():
        const = 0.5 * LOG2PI.to(theta.device) + torch.log(torch.as_tensor(float(self.sigma), device=theta.device))
        total = total - theta.numel() * const - 0.5 * torch.sum((theta / self.sigma) ** 2)
    return total


class OIVAE(object):
  def __init__(self, inference_model, generative_model, prior_z, prior_theta, lam, optimizers,
               device=None, process_group=None, data_parallel=True, device_eps=False):
    """Arguments as lib/oivae.py:26-40.  Extensions: `device`, `process_group` / `data_parallel` (one process per GPU:
    gradients are averaged over the ranks before the optimizers and the proximal step) and `device_eps`: draw the
    reparameterisation noise on the GPU (its own generator, seeded from torch's CPU generator at the first step) instead
    of on the CPU as the reference's Normal.sample does -- another random stream, no host draw and no copy per step."""
    self.inference_model = inference_model
    self.generative_model = generative_model
    self.prior_z = prior_z
    self.prior_theta = prior_theta
    self.lam = lam
    self.optimizers = optimizers
    self.process_group = process_group
    self.data_parallel = data_parallel
    self.device_eps = device_eps
    # moves every parameter into the engine's flat device buffers (raises without a GPU)
    self.engine = adopt(inference_model, generative_model, device=device)
    self._fused_sgd = None
    self._fused_adam = None
    self._foreign = []     # caller-owned torch.optim instances: (optimizer, [(parameter, its gradient view), ...])
    grad_of = {id(t): gv for t, gv in self.engine.bound}
    for opt in optimizers:
      if isinstance(opt, _EngineOptimizer):
        if not opt.covers(self.engine):
          raise _lib.OivaeError("%s must hold exactly the parameter set the reference's scripts give it "
                                "(bars_data_oivae.py:233-244)" % type(opt).__name__)
        opt._attach(self.engine)
        if isinstance(opt, _EngineSGD):
          self._fused_sgd = opt
        else:
          self._fused_adam = opt
      else:
        ps = [p for g in opt.param_groups for p in g["params"]]
        self._foreign.append((opt, [(p, grad_of[id(p)]) for p in ps if id(p) in grad_of]))
    self._all_native = not self._foreign and self._fused_sgd is not None and self._fused_adam is not None
    self._prior_dev = None
    self._head_cache = {}
    self._eps_gen = None

  def _prior(self):
    if self._prior_dev is None:
      dev = self.engine.device
      self._prior_dev = (self.prior_z.mu.detach().to(dev, torch.float32).contiguous(),
                         self.prior_z.sigma.detach().to(dev, torch.float32).contiguous())
    return self._prior_dev

  def _draw_eps(self, B, mc_samples):
    K = self.engine.spec.K
    if not self.device_eps:
      # the reference's stream: one CPU draw of [B, K] per sample, in order (lib/distributions.py:36-38)
      return torch.stack([torch.randn(B, K) for _ in range(mc_samples)])
    if self._eps_gen is None:
      self._eps_gen = torch.Generator(device=self.engine.device)
      self._eps_gen.manual_seed(int(torch.randint(0, 2 ** 62, (1,)).item()))
    return torch.randn(mc_samples, B, K, device=self.engine.device, generator=self._eps_gen)

  def step(self, X, prox_step_size, mc_samples, eps=None):
    """One training step; same arguments, side effects and return dict as
    lib/oivae.py:42-90.  `eps` ([mc_samples, B, K] standard-normal draws) is an
    extension for reproducible tests; by default the draws come from the CPU
    default generator exactly as the reference's Normal.sample makes them.

    Host work per step is independent of the number of groups when both optimizers are the engine's (oi_vae_amd.optim):
    gradient views are bound once, nothing is looped over.  Caller-owned torch.optim instances keep working; their
    zero_grad / step walk their parameter lists as torch does."""
    eng = self.engine
    B = X.size(0)
    if eps is None:
      eps = self._draw_eps(B, mc_samples)
    c, sig_off = probe_encoder_head(self.inference_model, self._head_cache)
    pm, ps = self._prior()
    for opt, pairs in self._foreign:
      opt.zero_grad()              # as lib/oivae.py:73-74 (sets .grad to None on torch >= 2) ...
      for t, gv in pairs:
        t.grad = gv                # ... and the views are bound again
    dp = self.data_parallel and parallel.is_active(self.process_group)
    if not dp and eng.small_step_applies(B, len(eps)):
      return self._small_step(X, eps, c, sig_off, pm, ps, prox_step_size)
    sync = parallel.BucketedGradSync(eng, self.process_group) if dp else None
    out = eng.forward_backward(X.detach() if torch.is_tensor(X) else X, eps, c, sig_off=sig_off,
                               gen_sig_off=eng.gen_sig_off, prior_mu=pm, prior_sigma=ps,
                               sigma_theta=float(self.prior_theta.sigma), lam=float(self.lam), grad_sync=sync)
    scalars = out["scalars"].clone()
    fuse = self._fused_sgd is not None and self.lam > 0
    if self._fused_sgd is not None:
      self._fused_sgd.fused_with_prox = fuse
    if dp and self._all_native:
      # bucket by bucket: the decoder-side gradients (reduced under the encoder backward GEMM) are final first, and
      # their Adam, SGD and proximal step run while the encoder bucket is still being reduced
      a = self._fused_adam
      sync.wait_decoder()
      eng.adam_step(a._lr(), betas=a._uniform("betas"), eps=a._uniform("eps"), part="dec")
      eng.sgd_prox_step(self._fused_sgd._lr(), prox_step_size if fuse else 0.0, do_prox=fuse)
      sync.wait_encoder()
      eng.adam_step(a._lr(), betas=a._uniform("betas"), eps=a._uniform("eps"), part="enc")
      parallel.reduce_scalars_(scalars, self.process_group)
      return self._info(out, scalars)
    if dp:
      sync.wait()
      parallel.reduce_scalars_(scalars, self.process_group)
    for opt in self.optimizers:
      opt.step()
    if fuse:
      eng.sgd_prox_step(self._fused_sgd._lr(), prox_step_size, do_prox=True)
    elif self.lam > 0:
      self.generative_model.proximal_step(prox_step_size)
    return self._info(out, scalars)

  def _small_step(self, X, eps, c, sig_off, pm, ps, prox_step_size):
    """The scripts' own small models (bars, mocap) on one GPU: the whole step is ONE launch (csrc/small_step.hip) --
    with the engine's optimizers inside it, with caller-owned torch.optim instances behind it."""
    eng = self.engine
    common = dict(sig_off=sig_off, gen_sig_off=eng.gen_sig_off, prior_mu=pm, prior_sigma=ps,
                  sigma_theta=float(self.prior_theta.sigma), lam=float(self.lam))
    Xd = X.detach() if torch.is_tensor(X) else X
    if self._all_native:
      a = self._fused_adam
      out = eng.small_step(Xd, eps, c, prox_step_size, lr=a._lr(), Ws_lr=self._fused_sgd._lr(), betas=a._uniform("betas"),
                           adam_eps=a._uniform("eps"), do_prox=self.lam > 0, optimizers=True, **common)
      return self._info(out, out["scalars"].clone())
    out = eng.small_step(Xd, eps, c, prox_step_size, lr=0.0, Ws_lr=0.0, optimizers=False, **common)
    scalars = out["scalars"].clone()
    fuse = self._fused_sgd is not None and self.lam > 0
    if self._fused_sgd is not None:
      self._fused_sgd.fused_with_prox = fuse
    for opt in self.optimizers:
      opt.step()
    if fuse:
      eng.sgd_prox_step(self._fused_sgd._lr(), prox_step_size, do_prox=True)
    elif self.lam > 0:
      self.generative_model.proximal_step(prox_step_size)
    return self._info(out, scalars)

  def evaluate(self, X, mc_samples, eps=None):
    """Held-out evaluation: the terms of `step`'s return dict for a batch X with no
    gradient, optimizer or proximal side effect (the test log-likelihood that the
    reference leaves stubbed, meg_oi_vae.py:258-260).  Under data parallelism each
    rank passes its shard and the scalars are combined as in `step`."""
    eng = self.engine
    if eps is None:
      eps = self._draw_eps(X.size(0), mc_samples)
    c, sig_off = probe_encoder_head(self.inference_model, self._head_cache)
    pm, ps = self._prior()
    out = eng.forward_backward(X.detach() if torch.is_tensor(X) else X, eps, c, sig_off=sig_off,
                               gen_sig_off=eng.gen_sig_off, prior_mu=pm, prior_sigma=ps,
                               sigma_theta=float(self.prior_theta.sigma), lam=float(self.lam),
                               evaluate_only=True)
    scalars = out["scalars"].clone()
    if self.data_parallel and parallel.is_active(self.process_group):
      parallel.reduce_scalars_(scalars, self.process_group)
    return self._info(out, scalars)

  def heldout_loglik(self, X, mc_samples, eps=None):
    """Held-out log-likelihood term of a batch, E_q log p(x | z) averaged over the batch (`loglik_term` of `evaluate`),
    evaluated on THIS rank only -- for test sets every rank holds whole (no reduction over ranks)."""
    dp = self.data_parallel
    self.data_parallel = False
    try:
      return self.evaluate(X, mc_samples, eps)['loglik_term']
    finally:
      self.data_parallel = dp

  @staticmethod
  def _info(out, scalars):
    return {
      'q_z': Normal(out["mu_q"], out["sig_q"]),
      'z_kl': scalars[0],
      'z_sample': out["z"],
      'loglik_term': scalars[1],
      'logprob_theta': scalars[2],
      'logprob_W': scalars[3],
      'loss': scalars[4],
      'elbo': scalars[5],
    }
