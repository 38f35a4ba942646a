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
               device=None, process_group=None, data_parallel=True):
    self.inference_model = inference_model
    self.generative_model = generative_model
    self.prior_z = prior_z
    self.prior_theta = prior_theta
    self.lam = lam
    self.optimizers = optimizers
    self.process_group = process_group
    self.data_parallel = data_parallel
    # moves every parameter into the engine's flat device buffers (raises without a GPU)
    self.engine = adopt(inference_model, generative_model, device=device)
    self._fused_sgd = None
    for opt in optimizers:
      if isinstance(opt, _EngineOptimizer):
        if not opt.covers(self.engine):
          raise _lib.OivaeError("%s must hold exactly the parameter set the reference's scripts give it "
                                "(bars_data_oivae.py:233-244)" % type(opt).__name__)
        opt._attach(self.engine)
        if isinstance(opt, _EngineSGD):
          self._fused_sgd = opt
    self._prior_dev = None

  def _prior(self):
    if self._prior_dev is None:
      dev = self.engine.device
      self._prior_dev = (self.prior_z.mu.detach().to(dev, torch.float32).contiguous(),
                         self.prior_z.sigma.detach().to(dev, torch.float32).contiguous())
    return self._prior_dev

  def step(self, X, prox_step_size, mc_samples, eps=None):
    """One training step; same arguments, side effects and return dict as
    lib/oivae.py:42-90.  `eps` ([mc_samples, B, K] standard-normal draws) is an
    extension for reproducible tests; by default the draws come from the CPU
    default generator exactly as the reference's Normal.sample makes them."""
    eng = self.engine
    K = eng.spec.K
    B = X.size(0)
    if eps is None:
      eps = torch.stack([torch.randn(B, K) for _ in range(mc_samples)])
    c, sig_off = probe_encoder_head(self.inference_model)
    pm, ps = self._prior()
    for opt in self.optimizers:
      opt.zero_grad()
    dp = self.data_parallel and parallel.is_active(self.process_group)
    sync = parallel.BucketedGradSync(eng, self.process_group) if dp else None
    out = eng.forward_backward(X.detach() if torch.is_tensor(X) else X, eps, c, sig_off=sig_off,
                               gen_sig_off=eng.gen_sig_off, prior_mu=pm, prior_sigma=ps,
                               sigma_theta=float(self.prior_theta.sigma), lam=float(self.lam), grad_sync=sync)
    scalars = out["scalars"].clone()
    if dp:
      sync.wait()
      parallel.reduce_scalars_(scalars, self.process_group)
    for t, gv in eng.bound:
      t.grad = gv
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
      eps = torch.stack([torch.randn(X.size(0), eng.spec.K) for _ in range(mc_samples)])
    c, sig_off = probe_encoder_head(self.inference_model)
    pm, ps = self._prior()
    out = eng.forward_backward(X.detach() if torch.is_tensor(X) else X, eps, c, sig_off=sig_off,
                               gen_sig_off=eng.gen_sig_off, prior_mu=pm, prior_sigma=ps,
                               sigma_theta=float(self.prior_theta.sigma), lam=float(self.lam),
                               evaluate_only=True)
    scalars = out["scalars"].clone()
    if self.data_parallel and parallel.is_active(self.process_group):
      parallel.reduce_scalars_(scalars, self.process_group)
    return self._info(out, scalars)

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
