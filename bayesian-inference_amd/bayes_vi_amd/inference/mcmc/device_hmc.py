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
