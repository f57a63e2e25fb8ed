"""HMC transitions that stay on the GPU stream: momenta, leapfrog updates, posterior assembly
and the Metropolis step run in the kernels of libbvi_hip (bvi_hmc_begin / bvi_leapfrog_pre /
bvi_leapfrog_post / bvi_hmc_end); per leapfrog step the host only enqueues the fused
likelihood pass (and, when rows are sharded over GPUs, the all-reduce of its partial sums)
between `pre` and `post`.  Same algorithm and interface as
`samplers.HamiltonianMonteCarloKernel` (tfp.mcmc.HamiltonianMonteCarlo of
bayes_vi/inference/mcmc/transition_kernels.py:122-128); random numbers are counter-based
Philox draws keyed by (seed; transition, chain, element), identical on every rank.
"""
import torch

from ... import _native


class DeviceHamiltonianMonteCarloKernel:
    trace_names = ('is_accepted', 'target_log_prob', 'step_size')

    def __init__(self, posterior, step_size=0.01, num_leapfrog_steps=5, seed=0):
        self.posterior = posterior
        self.blocks = posterior.device_blocks()
        if self.blocks is None:
            raise _native.BviError('this model has priors/bijectors outside the device-side posterior assembly')
        self.nb = len(self.blocks)
        self.step_size = float(step_size)
        self.num_leapfrog_steps = int(num_leapfrog_steps)
        self.seed = int(seed)
        self.transition = 0
        self._bufs = None

    def _buffers(self, z):
        c, pu = z.shape
        p = self.posterior.n_constrained
        if self._bufs is None or self._bufs['p'].shape != z.shape:
            f = lambda *s: torch.empty(*s, dtype=torch.float32, device=z.device)
            self._bufs = dict(p=f(c, pu), z0=f(c, pu), g0=f(c, pu), lp0=f(c), k0=f(c), theta=f(c, p),
                              acc=torch.empty(c, dtype=torch.uint8, device=z.device), lr=f(c))
        return self._bufs

    def bootstrap(self, z):
        lp, g = self.posterior.unconstrained_log_prob_and_grad(z, True)
        return {'target_log_prob': lp.contiguous(), 'grad': g.contiguous()}

    def one_step(self, z, state, generator=None, momenta=None, log_uniform=None):
        lib, ptr, st = _native.lib(), _native._ptr, _native._stream()
        z = z.clone()                       # the kernels update position, gradient and log-prob in place
        lp = state['target_log_prob'].clone()
        g = state['grad'].clone()
        c, pu = z.shape
        b = self._buffers(z)
        seed = self.seed if generator is None else (self.seed ^ int(generator.initial_seed()))
        step = self.transition
        self.transition += 1
        eps = self.step_size
        _native.check(lib.bvi_hmc_begin(c, pu, seed, step, ptr(z), ptr(g), ptr(lp),
                                        ptr(momenta.contiguous()) if momenta is not None else None,
                                        ptr(b['p']), ptr(b['z0']), ptr(b['g0']), ptr(b['lp0']), ptr(b['k0']), st),
                      'bvi_hmc_begin')
        lik = self.posterior.likelihood
        for _ in range(self.num_leapfrog_steps):
            _native.check(lib.bvi_leapfrog_pre(self.blocks, self.nb, c, eps, ptr(z), ptr(b['p']), ptr(g),
                                               ptr(b['theta']), st), 'bvi_leapfrog_pre')
            ll, gl = lik.logp_grad(b['theta'], True)          # fused data pass (+ all-reduce over row shards)
            _native.check(lib.bvi_leapfrog_post(self.blocks, self.nb, c, eps, ptr(z), ptr(b['p']), ptr(b['theta']),
                                                ptr(ll), ptr(gl), ptr(lp), ptr(g), st), 'bvi_leapfrog_post')
        _native.check(lib.bvi_hmc_end(c, pu, seed, step,
                                      ptr(log_uniform.contiguous()) if log_uniform is not None else None,
                                      ptr(z), ptr(g), ptr(lp), ptr(b['p']), ptr(b['z0']), ptr(b['g0']),
                                      ptr(b['lp0']), ptr(b['k0']), ptr(b['acc']), ptr(b['lr']), st), 'bvi_hmc_end')
        acc = b['acc'].to(torch.bool)
        trace = {'is_accepted': acc, 'target_log_prob': lp, 'log_accept_ratio': b['lr'].clone(),
                 'step_size': torch.full((), eps, dtype=z.dtype)}
        return z, {'target_log_prob': lp, 'grad': g}, trace


FUSED_MAX_ROWS = 1 << 16     # above this the batched kernels + per-step launches win


def fused_family(posterior):
    """(family id, X or None, y, n, d, k) when the whole-transition kernel (bvi_hmc_fused) covers this
    posterior -- an unsharded Normal / small regression / small 1-D mixture whose parameter layout
    is the family's own, built-in priors and bijectors, few enough rows -- else None."""
    from ...fused_target import NormalIID, LinearRegression, GaussianMixture1D
    lik = posterior.likelihood
    if getattr(lik, 'inner', None) is not None:
        if not getattr(lik, 'identity', False):
            return None
        lik = lik.inner
    if getattr(lik, 'sharded', False) or lik.device.type != 'cuda' or posterior.device_blocks() is None:
        return None
    if getattr(lik, 'n', FUSED_MAX_ROWS + 1) > FUSED_MAX_ROWS:
        return None
    if isinstance(lik, NormalIID):
        return (_native.FAMILY_NORMAL_IID, None, lik.y, lik.n, 0, 0)
    if isinstance(lik, LinearRegression) and lik.d <= 16:
        return (_native.FAMILY_LINREG, lik.X, lik.y, lik.n, lik.d, 0)
    if isinstance(lik, GaussianMixture1D) and lik.k <= 8:
        return (_native.FAMILY_GMM1D, None, lik.y, lik.n, 0, lik.k)
    return None


class FusedHamiltonianMonteCarloKernel:
    """HMC with whole transitions -- and whole runs of transitions -- inside one kernel launch
    (bvi_hmc_fused): position, momentum and gradient never leave the registers between the first
    and the last transition of a launch.  Same transition and random numbers as
    `DeviceHamiltonianMonteCarloKernel`; `many_steps` is what `sample_chain` uses once no step-size
    adaptation needs the host between transitions."""
    trace_names = ('is_accepted', 'target_log_prob', 'step_size')
    supports_many_steps = True

    def __init__(self, posterior, family, step_size=0.01, num_leapfrog_steps=5, seed=0):
        self.posterior = posterior
        self.family = family
        self.blocks = posterior.device_blocks()
        self.nb = len(self.blocks)
        self.step_size = float(step_size)
        self.num_leapfrog_steps = int(num_leapfrog_steps)
        self.seed = int(seed)
        self.transition = 0

    def _launch(self, z, g, lp, n_transitions, evaluate_initial, seed, momenta=None, log_uniform=None, outs=None):
        fam, X, y, n, d, k = self.family
        ptr = _native._ptr
        c = z.shape[0]
        zo, lpo, acc, lr = outs if outs is not None else (None, None, None, None)
        _native.check(_native.lib().bvi_hmc_fused(
            fam, ptr(X), ptr(y), n, d, k, self.blocks, self.nb, c, self.step_size, None, self.num_leapfrog_steps,
            n_transitions, seed, self.transition, 1 if evaluate_initial else 0, ptr(z), ptr(g), ptr(lp),
            ptr(momenta.contiguous()) if momenta is not None else None,
            ptr(log_uniform.contiguous()) if log_uniform is not None else None,
            ptr(zo), ptr(lpo), ptr(acc), ptr(lr), _native._stream()), 'bvi_hmc_fused')
        self.transition += n_transitions

    def bootstrap(self, z):
        z = z.contiguous().to(torch.float32)
        g = torch.empty_like(z)
        lp = torch.empty(z.shape[0], dtype=torch.float32, device=z.device)
        self._launch(z, g, lp, 0, True, self.seed)
        return {'target_log_prob': lp, 'grad': g}

    def many_steps(self, z, state, n, generator=None, momenta=None, log_uniform=None):
        """n transitions in one launch -> (zs[n, C, Pu], z, state, traces with a leading n axis)."""
        z = z.clone().contiguous()
        lp = state['target_log_prob'].clone()
        g = state['grad'].clone()
        c, pu = z.shape
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=z.device)
        zs, lps, lrs = f(n, c, pu), f(n, c), f(n, c)
        acc = torch.empty(n, c, dtype=torch.uint8, device=z.device)
        seed = self.seed if generator is None else (self.seed ^ int(generator.initial_seed()))
        self._launch(z, g, lp, n, False, seed, momenta, log_uniform, (zs, lps, acc, lrs))
        traces = {'is_accepted': acc.to(torch.bool), 'target_log_prob': lps, 'log_accept_ratio': lrs,
                  'step_size': torch.full((n,), self.step_size, dtype=z.dtype)}
        return zs, z, {'target_log_prob': lp, 'grad': g}, traces

    def one_step(self, z, state, generator=None, momenta=None, log_uniform=None):
        zs, z, state, tr = self.many_steps(z, state, 1, generator,
                                           momenta[None] if momenta is not None else None,
                                           log_uniform[None] if log_uniform is not None else None)
        return z, state, {k: v[0] for k, v in tr.items()}


class FusedNoUTurnSamplerKernel:
    """NUTS with whole transitions -- and whole runs of transitions -- inside one kernel launch
    (bvi_nuts_fused), one workgroup per chain.  Same algorithm and trace fields as
    `samplers.NoUTurnSamplerKernel`; random numbers are Philox draws keyed by (seed; transition,
    chain), so a run is reproducible from its seed."""
    trace_names = ('is_accepted', 'target_log_prob', 'log_accept_ratio', 'step_size',
                   'leapfrogs_taken', 'has_divergence', 'energy')
    supports_many_steps = True

    def __init__(self, posterior, family, step_size=0.01, max_tree_depth=5, max_energy_diff=1000.0, seed=0):
        self.posterior = posterior
        self.family = family
        self.blocks = posterior.device_blocks()
        self.nb = len(self.blocks)
        self.step_size = float(step_size)
        self.max_tree_depth = int(max_tree_depth)
        self.max_energy_diff = float(max_energy_diff)
        self.seed = int(seed)
        self.transition = 0

    def _launch(self, z, g, lp, n_transitions, evaluate_initial, seed, outs=None):
        fam, X, y, n, d, k = self.family
        ptr = _native._ptr
        outs = outs if outs is not None else (None,) * 6
        _native.check(_native.lib().bvi_nuts_fused(
            fam, ptr(X), ptr(y), n, d, k, self.blocks, self.nb, z.shape[0], self.step_size, None,
            self.max_tree_depth, self.max_energy_diff, n_transitions, seed, self.transition,
            1 if evaluate_initial else 0, ptr(z), ptr(g), ptr(lp), *[ptr(o) for o in outs], _native._stream()),
            'bvi_nuts_fused')
        self.transition += n_transitions

    def bootstrap(self, z):
        z = z.contiguous().to(torch.float32)
        g = torch.empty_like(z)
        lp = torch.empty(z.shape[0], dtype=torch.float32, device=z.device)
        self._launch(z, g, lp, 0, True, self.seed)
        return {'target_log_prob': lp, 'grad': g}

    def many_steps(self, z, state, n, generator=None):
        z = z.clone().contiguous()
        lp = state['target_log_prob'].clone()
        g = state['grad'].clone()
        c, pu = z.shape
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=z.device)
        zs, lps, lrs = f(n, c, pu), f(n, c), f(n, c)
        moved = torch.empty(n, c, dtype=torch.uint8, device=z.device)
        div = torch.empty(n, c, dtype=torch.uint8, device=z.device)
        leaps = torch.empty(n, c, dtype=torch.int32, device=z.device)
        seed = self.seed if generator is None else (self.seed ^ int(generator.initial_seed()))
        self._launch(z, g, lp, n, False, seed, (zs, lps, moved, lrs, leaps, div))
        traces = {'is_accepted': moved.to(torch.bool), 'target_log_prob': lps, 'log_accept_ratio': lrs,
                  'step_size': torch.full((n,), self.step_size, dtype=z.dtype),
                  'leapfrogs_taken': leaps.to(torch.int64), 'has_divergence': div.to(torch.bool), 'energy': -lps}
        return zs, z, {'target_log_prob': lp, 'grad': g}, traces

    def one_step(self, z, state, generator=None):
        zs, z, state, tr = self.many_steps(z, state, 1, generator)
        return z, state, {k: v[0] for k, v in tr.items()}


class FusedRandomWalkMetropolisKernel:
    """Random-walk Metropolis with whole runs of transitions inside one kernel launch (bvi_rwm_fused);
    same transition as `samplers.RandomWalkMetropolisKernel` with the default N(0, scale^2) proposal."""
    trace_names = ('is_accepted',)
    supports_many_steps = True

    def __init__(self, posterior, family, scale=1.0, seed=0):
        self.posterior = posterior
        self.family = family
        self.blocks = posterior.device_blocks()
        self.nb = len(self.blocks)
        self.scale = float(scale)
        self.seed = int(seed)
        self.transition = 0

    def _launch(self, z, g, lp, n_transitions, evaluate_initial, seed, outs=None):
        fam, X, y, n, d, k = self.family
        ptr = _native._ptr
        outs = outs if outs is not None else (None,) * 4
        _native.check(_native.lib().bvi_rwm_fused(
            fam, ptr(X), ptr(y), n, d, k, self.blocks, self.nb, z.shape[0], self.scale, n_transitions, seed,
            self.transition, 1 if evaluate_initial else 0, ptr(z), ptr(g), ptr(lp), *[ptr(o) for o in outs],
            _native._stream()), 'bvi_rwm_fused')
        self.transition += n_transitions

    def bootstrap(self, z):
        z = z.contiguous().to(torch.float32)
        g = torch.empty_like(z)
        lp = torch.empty(z.shape[0], dtype=torch.float32, device=z.device)
        self._launch(z, g, lp, 0, True, self.seed)
        return {'target_log_prob': lp, 'grad': g}

    def many_steps(self, z, state, n, generator=None):
        z = z.clone().contiguous()
        lp = state['target_log_prob'].clone()
        g = state['grad'].clone()
        c, pu = z.shape
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=z.device)
        zs, lps, lrs = f(n, c, pu), f(n, c), f(n, c)
        acc = torch.empty(n, c, dtype=torch.uint8, device=z.device)
        seed = self.seed if generator is None else (self.seed ^ int(generator.initial_seed()))
        self._launch(z, g, lp, n, False, seed, (zs, lps, acc, lrs))
        traces = {'is_accepted': acc.to(torch.bool), 'target_log_prob': lps, 'log_accept_ratio': lrs}
        return zs, z, {'target_log_prob': lp, 'grad': g}, traces

    def one_step(self, z, state, generator=None):
        zs, z, state, tr = self.many_steps(z, state, 1, generator)
        return z, state, {k: v[0] for k, v in tr.items()}
