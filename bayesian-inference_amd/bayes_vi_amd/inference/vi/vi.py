"""`VI`: variational inference (mirror of bayes_vi/inference/vi/vi.py)."""
import numpy as np
import torch

from ... import _native
from ...fused_target import differentiable_log_prob, replicate_from_rank0
from ..inference import Inference


def kl_reverse(logu):
    """Reverse KL in log space: f(log u) = -log u (tfp.vi.kl_reverse)."""
    return -logu


class _AdamState:
    """The update of tf.optimizers.Adam (the optimizer `VI.fit` defaults to, vi.py:77) on torch parameters:
        m <- b1 m + (1 - b1) g;  v <- b2 v + (1 - b2) g^2
        lr_t = lr sqrt(1 - b2^t) / (1 - b1^t);  x <- x - lr_t m / (sqrt(v) + epsilon)
    (epsilon sits outside the bias correction, unlike torch.optim.Adam).  The step counter lives on the
    parameters' device and nothing here reads it back, so a step can be recorded in a HIP graph."""

    def __init__(self, params, lr, betas, eps):
        self.params = list(params)
        self.lr, (self.b1, self.b2), self.eps = lr, betas, eps
        dev = self.params[0].device if self.params else torch.device('cpu')
        self.t = torch.zeros((), dtype=torch.float64, device=dev)
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    def zero_grad(self, set_to_none=False):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self):
        self.t += 1
        lr_t = (self.lr * torch.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)).to(torch.float32)
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad
            m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            p.sub_(lr_t * m / (v.sqrt() + self.eps))

    def clip_and_step(self, clip_norm=1.0):
        """tf.clip_by_norm(grad, clip_norm) per variable (vi.py:70) followed by `step`.  On the GPU each variable is
        one launch of bvi_clip_adam_step (norm, clip, moments, update) instead of some seven element-wise kernels,
        which is most of a step's launches for a flow with a few dozen small variables."""
        live = [(p, m, v) for p, m, v in zip(self.params, self.m, self.v) if p.grad is not None]
        on_device = bool(live) and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
                                       and p.grad.is_contiguous() and p.grad.dtype == torch.float32 for p, _, _ in live)
        if not on_device:
            with torch.no_grad():
                for p, _, _ in live:     # without a host sync
                    p.grad.mul_(torch.clamp(clip_norm / (p.grad.norm() + 1e-30), max=1.0))
            self.step()
            return
        lib, ptr, st = _native.lib(), _native._ptr, _native._stream()
        with torch.no_grad():
            for i, (p, m, v) in enumerate(live):
                _native.check(lib.bvi_clip_adam_step(p.numel(), ptr(p), ptr(p.grad), ptr(m), ptr(v), ptr(self.t),
                                                     1 if i == 0 else 0, self.lr, self.b1, self.b2, self.eps,
                                                     float(clip_norm), st), 'bvi_clip_adam_step')


class Adam:
    """Optimizer description with the reference's spelling, tf.optimizers.Adam(learning_rate=...);
    bound to the surrogate's parameters when `VI.fit` starts."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate, self.betas, self.eps = learning_rate, (beta_1, beta_2), epsilon

    def bind(self, params, capturable=False):
        return _AdamState(params, self.learning_rate, self.betas, self.eps)


class VI(Inference):
    """Same construction and `fit` signature as the reference (vi.py:37, 77-78).

    The whole dataset is one batch on the GPU (vi.py:40); `target_log_prob` is the closure of
    vi.py:46-52 on flat constrained samples [S, P(+extra)], evaluated -- value AND gradient in the
    same pass -- by the fused kernels and exposed to autograd as a single node.
    """

    def __init__(self, model, dataset, surrogate_posterior, discrepancy_fn=kl_reverse, group=None, precision=None):
        super().__init__(model=model, dataset=dataset)
        self.features, self.targets = dataset.full_batch()
        self.distribution = model.get_joint_distribution(num_samples=self.targets.shape[0], features=self.features,
                                                         group=group, precision=precision)
        self.posterior = self.distribution.posterior(self.targets)
        self.device = self.posterior.likelihood.device
        self.surrogate_posterior = surrogate_posterior.to(self.device)
        self.group = group
        # rows sharded over ranks: every rank optimises the same surrogate on the same Monte-Carlo samples, so the
        # parameters and the device's random stream are taken from rank 0
        for p in self.surrogate_posterior.parameters():
            replicate_from_rank0(p.data, group)
        if getattr(self.posterior.likelihood, 'sharded', False) or \
                getattr(getattr(self.posterior.likelihood, 'inner', None), 'sharded', False):
            seed = replicate_from_rank0(torch.tensor([torch.initial_seed() % (1 << 62)], dtype=torch.int64), group)
            torch.manual_seed(int(seed[0]))
        p = self.model.flat_param_event_ndims
        base = lambda sample: differentiable_log_prob(self.posterior, sample[:, :p].contiguous())
        self.target_log_prob = self.surrogate_posterior.get_corrected_target_log_prob_fn(base)
        self.discrepancy_fn = discrepancy_fn

    def safe_discrepancy_fn(self, logu):
        """Discrepancy with non-finite entries dropped before the mean (vi.py:55-58)."""
        d = self.discrepancy_fn(logu)
        return d[torch.isfinite(d)]

    def loss(self, sample_size):
        """Monte-Carlo variational loss with reparameterised samples
        (tfp.vi.monte_carlo_variational_loss as called at vi.py:109-116).  The mean over the finite
        discrepancies (vi.py:55-58) is taken as a masked mean: same value and gradient as
        `safe_discrepancy_fn(logu).mean()`, without the data-dependent shape (and host round trip)
        of boolean indexing."""
        q, logq = self.surrogate_posterior.sample_and_log_prob(sample_size)
        logu = self.target_log_prob(q) - logq
        d = self.discrepancy_fn(logu)
        finite = torch.isfinite(d)
        return torch.where(finite, d, torch.zeros_like(d)).sum() / finite.sum()

    # ---- ADVI (mean-field or full-rank) with the built-in priors / bijectors: the whole step on the device, no autograd ----
    def _fused_advi_eligible(self, opt):
        from .surrogate_posteriors import ADVI
        sur = self.surrogate_posterior
        if not (getattr(self, 'fused_advi', True) and isinstance(sur, ADVI) and sur.extra_ndims == 0
                and self.discrepancy_fn is kl_reverse and isinstance(opt, _AdamState) and self.device.type == 'cuda'
                and self.posterior.device_blocks() is not None):
            return False
        scale = sur.raw_scale if sur.mean_field else sur.raw_tril
        return (sur.loc.requires_grad and scale.requires_grad and scale.is_contiguous()
                and [id(p) for p in opt.params] == [id(sur.loc), id(scale)])

    def _step_fused_advi(self, opt, sample_size):
        """One step of `_step` for the ADVI surrogates (mean-field, full-rank) when the model's priors and bijectors are the
        device-side ones: sample -> constrain -> fused data pass -> assemble -> closed-form ELBO gradients ->
        per-variable clip + Adam, eight launches (surrogate_posteriors.py:113-123, vi.py:55-75, 109-116).  Same
        loss, clipped gradients (left in `.grad`) and parameters as autograd over the same formulas."""
        lib, ptr, st = _native.lib(), _native._ptr, _native._stream()
        sur, post = self.surrogate_posterior, self.posterior
        s, pu = int(sample_size), sur.unconstrained_event_ndims
        eps = sur.base_noise(s, pu).contiguous()
        b = getattr(self, '_advi_bufs', None)
        if b is None or b['z'].shape != (s, pu):
            f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=self.device)
            b = self._advi_bufs = dict(z=f(s, pu), loss=f(()))
        scale = sur.raw_scale if sur.mean_field else sur.raw_tril
        sample, grads = ((lib.bvi_advi_meanfield_sample, lib.bvi_advi_meanfield_grads) if sur.mean_field
                         else (lib.bvi_advi_fullrank_sample, lib.bvi_advi_fullrank_grads))
        for prm in (sur.loc, scale):
            if prm.grad is None:
                prm.grad = torch.zeros_like(prm)
        with torch.no_grad():
            _native.check(sample(s, pu, ptr(eps), ptr(sur.loc), ptr(scale), ptr(b['z']), st), 'bvi_advi_*_sample')
            lp, gz = post.unconstrained_log_prob_and_grad(b['z'], True)
            _native.check(grads(s, pu, ptr(lp), ptr(gz), ptr(eps), ptr(scale), ptr(b['loss']), ptr(sur.loc.grad),
                                ptr(scale.grad), st), 'bvi_advi_*_grads')
            for i, (prm, m, v) in enumerate(zip(opt.params, opt.m, opt.v)):
                _native.check(lib.bvi_clip_adam_step(prm.numel(), ptr(prm), ptr(prm.grad), ptr(m), ptr(v), ptr(opt.t),
                                                     1 if i == 0 else 0, opt.lr, opt.b1, opt.b2, opt.eps, 1.0, st),
                              'bvi_clip_adam_step')
        return b['loss'].clone()

    def _step(self, opt, params, sample_size):
        if self._fused_advi_eligible(opt):
            return self._step_fused_advi(opt, sample_size)
        opt.zero_grad(set_to_none=False)
        loss = self.loss(sample_size)
        loss.backward()
        if isinstance(opt, _AdamState):
            opt.clip_and_step(1.0)
        else:
            for p in params:
                if p.grad is not None:     # tf.clip_by_norm(grad, 1.0), without a host sync
                    p.grad.mul_(torch.clamp(1.0 / (p.grad.norm() + 1e-30), max=1.0))
            opt.step()
        return loss.detach()

    def fit(self, optimizer=None, num_steps=10000, sample_size=1, num_samples_to_approx_marginals=10000,
            progress_bar=True, capture_graph=None):
        """Returns (approx_posterior, losses[num_steps]) (vi.py:77-127).  Every gradient is clipped
        to norm 1 per variable, not globally (vi.py:70).

        `capture_graph` (default: on for GPU runs with the built-in `Adam`): after three eager steps
        the whole step -- sampling, fused log-posterior pass, backward, clipping, optimizer update -- is
        captured once as a HIP graph and replayed, which removes the per-kernel launch cost that
        dominates on small datasets.  The steps and their results are the same either way.  (A caller that evaluated
        `loss()` itself beforehand must not keep that loss tensor alive into `fit`: its autograd graph pins the
        parameters' gradient accumulators to the stream it ran on, and torch then ties the capturing stream to it.)"""
        optimizer = optimizer if optimizer is not None else Adam()
        params = [p for p in self.surrogate_posterior.parameters() if p.requires_grad]
        on_gpu = self.device.type == 'cuda'
        if capture_graph is None:
            # a row-sharded likelihood has a collective inside the step: that stays eager unless asked for
            lik = self.posterior.likelihood
            sharded = getattr(lik, 'sharded', False) or getattr(getattr(lik, 'inner', None), 'sharded', False)
            capture_graph = on_gpu and hasattr(optimizer, 'bind') and num_steps > 8 and not sharded
        if capture_graph and not (on_gpu and hasattr(optimizer, 'bind')):
            raise ValueError('capture_graph needs a GPU run and the built-in Adam description')
        if hasattr(optimizer, 'bind'):
            opt = optimizer.bind(params, capturable=True) if capture_graph else optimizer.bind(params)
        else:
            opt = optimizer(params)
        losses_dev = torch.zeros(num_steps, dtype=torch.float32, device=self.device)
        bar = None
        if progress_bar:
            try:
                from tqdm import tqdm
                bar = tqdm(total=num_steps, leave=False)
            except Exception:
                bar = None
        n_eager = min(num_steps, 3) if capture_graph else num_steps
        for i in range(n_eager):
            losses_dev[i] = self._step(opt, params, sample_size)
            if bar is not None:
                bar.update(1)
        if capture_graph and num_steps > n_eager:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_loss = self._step(opt, params, sample_size)
            for i in range(n_eager, num_steps):
                graph.replay()
                losses_dev[i] = static_loss
                if bar is not None and (i & 63) == 0:
                    bar.update(64)
        if bar is not None:
            bar.close()
        losses = losses_dev.cpu().numpy().astype(np.float64)
        approx_posterior = self.surrogate_posterior.approx_joint_marginal_posteriors(num_samples_to_approx_marginals)
        return approx_posterior, losses

    def stochastic_fit(self, *args, **kwargs):
        raise NotImplementedError('minibatch fitting is outside the fused full-batch hot path')
