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

    def __init__(self, posterior, step_size=0.01, num_leapfrog_steps=5, seed=0, graph_replay=False):
        self.posterior = posterior
        self.blocks = posterior.device_blocks()
        if self.blocks is None:
            raise _native.BviError('this model has priors/bijectors outside the device-side posterior assembly')
        self.nb = len(self.blocks)
        self.step_size = float(step_size)
        self.num_leapfrog_steps = int(num_leapfrog_steps)
        self.seed = int(seed)
        self.transition = 0
        self.graph_replay = bool(graph_replay)
        self._bufs = None
        posterior.use_relative_log_prob()

    def _buffers(self, z):
        c, pu = z.shape
        p = self.posterior.n_constrained
        if self._bufs is None or self._bufs['p'].shape != z.shape:
            f = lambda *s: torch.empty(*s, dtype=torch.float32, device=z.device)
            self._bufs = dict(p=f(c, pu), z0=f(c, pu), g0=f(c, pu), lp0=f(c), k0=f(c), theta=f(c, p),
                              acc=torch.empty(c, dtype=torch.uint8, device=z.device), lr=f(c), lp_abs=f(c))
        return self._bufs

    def bootstrap(self, z):
        lp, g = self.posterior.initial_log_prob_and_grad(z, True)
        return dict(self.posterior.log_prob_state(lp.contiguous()), grad=g.contiguous())

    def _transition(self, z, g, lp, b, seed, step, momenta=None, log_uniform=None, step_offset=None):
        """One transition in place on (z, g, lp); accept flags and log ratio land in b['acc'], b['lr'], the
        absolute log-densities in b['lp_abs'] (lp itself is kept relative to a level that follows the chains)."""
        lib, ptr, st = _native.lib(), _native._ptr, _native._stream()
        c, pu = z.shape
        eps = self.step_size
        _native.check(lib.bvi_hmc_begin(c, pu, seed, step, ptr(z), ptr(g), ptr(lp),
                                        ptr(momenta.contiguous()) if momenta is not None else None,
                                        ptr(b['p']), ptr(b['z0']), ptr(b['g0']), ptr(b['lp0']), ptr(b['k0']),
                                        ptr(step_offset), st), 'bvi_hmc_begin')
        lik = self.posterior.likelihood
        _native.check(lib.bvi_leapfrog_pre(self.blocks, self.nb, c, eps, ptr(z), ptr(b['p']), ptr(g),
                                           ptr(b['theta']), st), 'bvi_leapfrog_pre')
        for i in range(self.num_leapfrog_steps):
            ll, gl = lik.logp_grad(b['theta'], True, relative=True)          # fused data pass (+ all-reduce over row shards)
            # interior steps: the second half of this step and the first half of the next in one launch
            last = i + 1 == self.num_leapfrog_steps
            fn, name = (lib.bvi_leapfrog_post, 'bvi_leapfrog_post') if last else (lib.bvi_leapfrog_mid, 'bvi_leapfrog_mid')
            _native.check(fn(self.blocks, self.nb, c, eps, ptr(z), ptr(b['p']), ptr(b['theta']), ptr(ll), ptr(gl),
                             ptr(lp), ptr(g), st), name)
        _native.check(lib.bvi_hmc_end(c, pu, seed, step,
                                      ptr(log_uniform.contiguous()) if log_uniform is not None else None,
                                      ptr(z), ptr(g), ptr(lp), ptr(b['p']), ptr(b['z0']), ptr(b['g0']),
                                      ptr(b['lp0']), ptr(b['k0']), ptr(b['acc']), ptr(b['lr']), ptr(step_offset), st),
                      'bvi_hmc_end')
        self.posterior.rebase(lp, b['lp_abs'])

    def one_step(self, z, state, generator=None, momenta=None, log_uniform=None):
        z = z.clone().contiguous()          # the kernels update position, gradient and log-prob in place
        lp = self.posterior.state_log_prob(state)
        g = state['grad'].clone()
        b = self._buffers(z)
        seed = self.seed if generator is None else (self.seed ^ int(generator.initial_seed()))
        step = self.transition
        self.transition += 1
        self._transition(z, g, lp, b, seed, step, momenta, log_uniform)
        acc = b['acc'].to(torch.bool)
        lp_abs = b['lp_abs'].clone()
        trace = {'is_accepted': acc, 'target_log_prob': lp_abs, 'log_accept_ratio': b['lr'].clone(),
                 'step_size': torch.full((), self.step_size, dtype=z.dtype)}
        return z, dict(self.posterior.log_prob_state(lp, lp_abs), grad=g), trace

    supports_many_steps = True
    GRAPH_MIN_TRANSITIONS = 8

    def many_steps(self, z, state, n, generator=None):
        """n transitions -> (zs[n, C, Pu], z, state, traces with a leading n axis), enqueued back to back
        with no host synchronisation; every transition copies its results into row t of the outputs on the
        stream.  With `graph_replay` one transition -- begin, L x (pre, data pass, post), end, the copies --
        is recorded once as a HIP graph and replayed; the transition number the Philox draws are keyed by
        lives in a device word that each transition advances, so the chain is the same either way (and the
        one `one_step` produces).  Replay only pays when the host cannot keep up with the GPU: measured on
        an otherwise idle host it is slower than plain launches (77 against 44 us per leapfrog step at
        N = 200k, D = 32, 64 chains), hence off by default.  A row-sharded likelihood (a collective per
        leapfrog step) is never captured."""
        lik = self.posterior.likelihood
        sharded = getattr(lik, 'sharded', False) or getattr(getattr(lik, 'inner', None), 'sharded', False)
        z = z.clone().contiguous()
        lp = self.posterior.state_log_prob(state)
        g = state['grad'].clone()
        c, pu = z.shape
        b = self._buffers(z)
        dev = z.device
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        zs, lps, lrs = f(n, c, pu), f(n, c), f(n, c)
        accs = torch.empty(n, c, dtype=torch.uint8, device=dev)
        seed = self.seed if generator is None else (self.seed ^ int(generator.initial_seed()))
        base = self.transition
        self.transition += n
        idx = torch.zeros(1, dtype=torch.int64, device=dev)     # transitions done; also the row to write next

        def transition_and_record(step_offset):
            self._transition(z, g, lp, b, seed, base, step_offset=step_offset)
            zs.index_copy_(0, idx, z.unsqueeze(0))
            lps.index_copy_(0, idx, b['lp_abs'].unsqueeze(0))
            lrs.index_copy_(0, idx, b['lr'].unsqueeze(0))
            accs.index_copy_(0, idx, b['acc'].unsqueeze(0))
            idx.add_(1)

        if not self.graph_replay or sharded or n < self.GRAPH_MIN_TRANSITIONS:
            for _ in range(n):
                transition_and_record(idx)
        else:
            transition_and_record(idx)          # eager: sizes every workspace before the capture
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                transition_and_record(idx)
            for _ in range(n - 1):
                graph.replay()
        traces = {'is_accepted': accs.to(torch.bool), 'target_log_prob': lps, 'log_accept_ratio': lrs,
                  'step_size': torch.full((n,), self.step_size, dtype=z.dtype)}
        return zs, z, dict(self.posterior.log_prob_state(lp), grad=g), traces


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
        outs = outs if outs is not None else (None,) * 7
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
        zs, lps, lrs, energy = f(n, c, pu), f(n, c), f(n, c), f(n, c)
        moved = torch.empty(n, c, dtype=torch.uint8, device=z.device)
        div = torch.empty(n, c, dtype=torch.uint8, device=z.device)
        leaps = torch.empty(n, c, dtype=torch.int32, device=z.device)
        seed = self.seed if generator is None else (self.seed ^ int(generator.initial_seed()))
        self._launch(z, g, lp, n, False, seed, (zs, lps, moved, lrs, leaps, div, energy))
        traces = {'is_accepted': moved.to(torch.bool), 'target_log_prob': lps, 'log_accept_ratio': lrs,
                  'step_size': torch.full((n,), self.step_size, dtype=z.dtype),
                  'leapfrogs_taken': leaps.to(torch.int64), 'has_divergence': div.to(torch.bool), 'energy': energy}
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


class BatchedDeviceNoUTurnSamplerKernel:
    """NUTS for datasets that need the batched likelihood kernels (any size, row shards over GPUs):
    the tree bookkeeping of every chain runs in the bvi_nuts_* kernels, the leapfrog halves in
    bvi_leapfrog_{pre,post}_chainwise with one signed step size per chain, and per leapfrog step the
    host enqueues five launches (pre, fused data pass + finalize, post, leaf) and reads back one word.
    Chains advance in lock-step, so one transition costs (max over chains of leapfrogs) data passes.
    Same algorithm, trace fields and Philox draws as `FusedNoUTurnSamplerKernel`
    (tfp.mcmc.NoUTurnSampler of bayes_vi/inference/mcmc/transition_kernels.py:238-246)."""
    trace_names = ('is_accepted', 'target_log_prob', 'log_accept_ratio', 'step_size',
                   'leapfrogs_taken', 'has_divergence', 'energy')

    def __init__(self, posterior, step_size=0.01, max_tree_depth=5, max_energy_diff=1000.0, seed=0):
        self.posterior = posterior
        self.blocks = posterior.device_blocks()
        if self.blocks is None:
            raise _native.BviError('this model has priors/bijectors outside the device-side posterior assembly')
        if not 1 <= int(max_tree_depth) <= 10:
            raise _native.BviError('max_tree_depth must be in 1..10 on the device path')
        self.nb = len(self.blocks)
        self.step_size = float(step_size)
        self.max_tree_depth = int(max_tree_depth)
        self.max_energy_diff = float(max_energy_diff)
        self.seed = int(seed)
        self.transition = 0
        self._bufs = None
        posterior.use_relative_log_prob()

    def _buffers(self, c, pu, device):
        if self._bufs is None or self._bufs['q'].shape != (c, pu):
            f = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
            i = lambda *s: torch.empty(*s, dtype=torch.int32, device=device)
            b = {n: f(c, pu) for n in ('q', 'p', 'gq', 'ql', 'pl', 'gl', 'qr', 'pr', 'gr', 'sub_q', 'sub_g')}
            b.update({n: f(c) for n in ('lpq', 'sub_lp', 'h0', 'log_w', 'sub_log_w', 'sum_acc', 'log_u_sub', 'se', 'energy',
                                         'sub_energy')})
            b.update({n: i(c) for n in ('n_leap', 'leaf_ctr', 'flags')})
            b['ck_q'], b['ck_p'] = f(self.max_tree_depth + 1, c, pu), f(self.max_tree_depth + 1, c, pu)
            b['theta'] = f(c, self.posterior.n_constrained)
            b['any'] = i((1 << self.max_tree_depth) + self.max_tree_depth)
            self._bufs = b
        return self._bufs

    def bootstrap(self, z):
        lp, g = self.posterior.initial_log_prob_and_grad(z, True)
        return dict(self.posterior.log_prob_state(lp.contiguous()), grad=g.contiguous())

    def one_step(self, z, state, generator=None):
        lib, ptr, st = _native.lib(), _native._ptr, _native._stream()
        z = z.clone().contiguous()          # the kernels update position, gradient and log-prob in place
        lp = self.posterior.state_log_prob(state)
        g = state['grad'].clone()
        c, pu = z.shape
        b = self._buffers(c, pu, z.device)
        seed = self.seed if generator is None else (self.seed ^ int(generator.initial_seed()))
        step = self.transition
        self.transition += 1
        s = _native.NutsState(n_chains=c, n_unconstrained=pu, max_tree_depth=self.max_tree_depth,
                              step_size=self.step_size, max_energy_diff=self.max_energy_diff,
                              z=z.data_ptr(), lp=lp.data_ptr(), grad=g.data_ptr(),
                              **{n: b[n].data_ptr() for n in _native._NUTS_FLOAT_ARRAYS + _native._NUTS_INT_ARRAYS
                                 if n not in ('z', 'lp', 'grad')})
        flag, nflag = b['any'], 0
        flag.zero_()
        flag_ptr = flag.data_ptr()          # word n of `flag` is passed as an address: no tensor view per launch
        _native.check(lib.bvi_nuts_begin(s, seed, step, st), 'bvi_nuts_begin')
        lik = self.posterior.likelihood
        for depth in range(self.max_tree_depth):
            _native.check(lib.bvi_nuts_subtree_begin(s, seed, step, depth, st), 'bvi_nuts_subtree_begin')
            n_leaves = 1 << depth
            for leaf in range(n_leaves):
                _native.check(lib.bvi_leapfrog_pre_chainwise(self.blocks, self.nb, c, ptr(b['se']), ptr(b['q']),
                                                             ptr(b['p']), ptr(b['gq']), ptr(b['theta']), st),
                              'bvi_leapfrog_pre_chainwise')
                ll, gl = lik.logp_grad(b['theta'], True, relative=True)      # fused data pass (+ all-reduce over row shards)
                _native.check(lib.bvi_leapfrog_post_chainwise(self.blocks, self.nb, c, ptr(b['se']), ptr(b['q']),
                                                              ptr(b['p']), ptr(b['theta']), ptr(ll), ptr(gl),
                                                              ptr(b['lpq']), ptr(b['gq']), st),
                              'bvi_leapfrog_post_chainwise')
                _native.check(lib.bvi_nuts_leaf(s, seed, step, leaf, flag_ptr + 4 * nflag, st), 'bvi_nuts_leaf')
                nflag += 1
                # The host only needs to know when NO chain is building any more.  U-turns inside a subtree
                # are found at odd leaves, so that is when the word is read back (a divergence at an even
                # leaf is noticed one frozen leapfrog late); after the last leaf the subtree ends either way.
                if (leaf & 1) and leaf + 1 < n_leaves and int(flag[nflag - 1].item()) == 0:
                    break
            _native.check(lib.bvi_nuts_subtree_end(s, flag_ptr + 4 * nflag, st), 'bvi_nuts_subtree_end')
            nflag += 1
            if depth + 1 < self.max_tree_depth and int(flag[nflag - 1].item()) == 0:
                break
        fl = b['flags']
        n_leap = b['n_leap'].to(torch.int64)
        lr = torch.log(torch.clamp(b['sum_acc'] / torch.clamp(n_leap, min=1).to(z.dtype), min=1e-30))
        energy = self.posterior.absolute(b['energy'])      # stored relative to the same level as lp
        lp_abs = torch.empty_like(lp)
        self.posterior.rebase(lp, lp_abs)   # lp stays relative to a level that follows the chains
        trace = {'is_accepted': (fl & _native.NUTS_MOVED) != 0, 'target_log_prob': lp_abs, 'log_accept_ratio': lr,
                 'step_size': torch.full((), self.step_size, dtype=z.dtype), 'leapfrogs_taken': n_leap,
                 'has_divergence': (fl & _native.NUTS_DIVERGED) != 0, 'energy': energy}
        return z, dict(self.posterior.log_prob_state(lp, lp_abs), grad=g), trace


class DeviceRandomWalkMetropolisKernel:
    """Random-walk Metropolis (tfp.mcmc.RandomWalkMetropolis with the default N(0, scale^2) proposal,
    bayes_vi/inference/mcmc/transition_kernels.py:319-321) for datasets beyond the one-workgroup-per-chain
    kernel, composed from the HMC entry points: bvi_hmc_begin saves the state and draws the step p,
    bvi_leapfrog_pre with a zero gradient and step size `scale` moves z += scale p and constrains,
    the value-only data pass and bvi_posterior_assemble give log pi at the proposal, and bvi_hmc_end
    accepts on log pi' - log pi (p is unchanged, so its kinetic terms cancel) and restores rejected chains.
    Same Philox draws as `FusedRandomWalkMetropolisKernel`; nothing synchronises with the host."""
    trace_names = ('is_accepted',)
    supports_many_steps = True

    def __init__(self, posterior, scale=1.0, seed=0):
        self.posterior = posterior
        self.blocks = posterior.device_blocks()
        if self.blocks is None:
            raise _native.BviError('this model has priors/bijectors outside the device-side posterior assembly')
        self.nb = len(self.blocks)
        self.scale = float(scale)
        self.seed = int(seed)
        self.transition = 0
        self._bufs = None
        posterior.use_relative_log_prob()

    def _buffers(self, z):
        c, pu = z.shape
        if self._bufs is None or self._bufs['p'].shape != z.shape:
            f = lambda *s: torch.empty(*s, dtype=torch.float32, device=z.device)
            self._bufs = dict(p=f(c, pu), z0=f(c, pu), g0=f(c, pu), lp0=f(c), k0=f(c), lr=f(c),
                              gzero=torch.zeros(c, pu, dtype=torch.float32, device=z.device),
                              theta=f(c, self.posterior.n_constrained),
                              acc=torch.empty(c, dtype=torch.uint8, device=z.device))
        return self._bufs

    def bootstrap(self, z):
        lp, _ = self.posterior.initial_log_prob_and_grad(z, False)
        return self.posterior.log_prob_state(lp.contiguous())

    def _transition(self, z, lp, b, seed, step):
        lib, ptr, st = _native.lib(), _native._ptr, _native._stream()
        c, pu = z.shape
        g = b['gzero']
        _native.check(lib.bvi_hmc_begin(c, pu, seed, step, ptr(z), ptr(g), ptr(lp), None, ptr(b['p']), ptr(b['z0']),
                                        ptr(b['g0']), ptr(b['lp0']), ptr(b['k0']), None, st), 'bvi_hmc_begin')
        _native.check(lib.bvi_leapfrog_pre(self.blocks, self.nb, c, self.scale, ptr(z), ptr(b['p']), ptr(g),
                                           ptr(b['theta']), st), 'bvi_leapfrog_pre')
        ll, _ = self.posterior.likelihood.logp_grad(b['theta'], False, relative=True)
        _native.check(lib.bvi_posterior_assemble(self.blocks, self.nb, c, ptr(z), ptr(b['theta']), ptr(ll), None,
                                                 ptr(lp), None, st), 'bvi_posterior_assemble')
        _native.check(lib.bvi_hmc_end(c, pu, seed, step, None, ptr(z), ptr(g), ptr(lp), ptr(b['p']), ptr(b['z0']),
                                      ptr(b['g0']), ptr(b['lp0']), ptr(b['k0']), ptr(b['acc']), ptr(b['lr']), None,
                                      st), 'bvi_hmc_end')
        self.posterior.rebase(lp)           # lp stays relative to a level that follows the chains

    def one_step(self, z, state, generator=None):
        z = z.clone().contiguous()
        lp = self.posterior.state_log_prob(state)
        b = self._buffers(z)
        seed = self.seed if generator is None else (self.seed ^ int(generator.initial_seed()))
        self._transition(z, lp, b, seed, self.transition)
        self.transition += 1
        return z, self.posterior.log_prob_state(lp), {'is_accepted': b['acc'].to(torch.bool),
                                                      'log_accept_ratio': b['lr'].clone()}

    def many_steps(self, z, state, n, generator=None):
        z = z.clone().contiguous()
        lp = self.posterior.state_log_prob(state)
        c, pu = z.shape
        b = self._buffers(z)
        zs = torch.empty(n, c, pu, dtype=torch.float32, device=z.device)
        accs = torch.empty(n, c, dtype=torch.uint8, device=z.device)
        lrs = torch.empty(n, c, dtype=torch.float32, device=z.device)
        seed = self.seed if generator is None else (self.seed ^ int(generator.initial_seed()))
        for t in range(n):
            self._transition(z, lp, b, seed, self.transition + t)
            zs[t].copy_(z)
            accs[t].copy_(b['acc'])
            lrs[t].copy_(b['lr'])
        self.transition += n
        return zs, z, self.posterior.log_prob_state(lp), {'is_accepted': accs.to(torch.bool), 'log_accept_ratio': lrs}

