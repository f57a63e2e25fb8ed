"""Batched Markov transition kernels on torch tensors.

The reference delegates these to TensorFlow-Probability:
`tfp.mcmc.HamiltonianMonteCarlo`, `tfp.mcmc.NoUTurnSampler`,
`tfp.mcmc.RandomWalkMetropolis`, `tfp.mcmc.SimpleStepSizeAdaptation`,
`tfp.mcmc.DualAveragingStepSizeAdaptation`, driven by `tfp.mcmc.sample_chain`
(bayes_vi/inference/mcmc/transition_kernels.py:122-128, 238-246, 319-321;
stepsize_adaptation_kernels.py:50-58, 90-101; mcmc.py:187-198).  TFP is not part
of this build, so the kernels are written from the published algorithms
(Metropolis; Neal's HMC; Hoffman & Gelman's NUTS with multinomial sampling and
their dual-averaging recursion) with TFP's defaults and trace fields.  Exact
random streams differ from TFP's, so agreement is statistical (posterior
moments within Monte-Carlo error).

Everything works on a flat unconstrained state z[C, P] (chains on the leading
axis, lock-step across chains) and a callable
    target(z, want_grad) -> (log_prob[C], grad[C, P] or None)
which on the GPU is `Posterior.unconstrained_log_prob_and_grad`: one fused pass
over the dataset per call.  Nothing here depends on the device.
"""
from __future__ import annotations

import math

import torch


def _finite_or_neg_inf(x):
    return torch.where(torch.isfinite(x), x, torch.full_like(x, -math.inf))


class RandomWalkMetropolisKernel:
    """q' = q + scale * N(0, I); accept with probability min(1, pi(q')/pi(q)).
    TFP default proposal: random_walk_normal_fn(scale=1.)."""
    trace_names = ('is_accepted',)

    def __init__(self, target, scale=1.0, new_state_fn=None):
        self.target, self.scale, self.new_state_fn = target, scale, new_state_fn

    def bootstrap(self, z):
        lp, _ = self.target(z, False)
        return {'target_log_prob': lp}

    def one_step(self, z, state, generator=None):
        if self.new_state_fn is not None:
            prop = self.new_state_fn(z, generator)
        else:
            prop = z + self.scale * torch.randn(z.shape, generator=generator, device=z.device, dtype=z.dtype)
        lp1, _ = self.target(prop, False)
        log_ratio = _finite_or_neg_inf(lp1 - state['target_log_prob'])
        log_u = torch.log(torch.rand(z.shape[0], generator=generator, device=z.device, dtype=z.dtype))
        acc = log_u < log_ratio
        z = torch.where(acc[:, None], prop, z)
        state = {'target_log_prob': torch.where(acc, lp1, state['target_log_prob'])}
        return z, state, {'is_accepted': acc}


class HamiltonianMonteCarloKernel:
    """Neal's HMC: p ~ N(0, I); L leapfrog steps on grad log pi; Metropolis accept on the
    energy difference; non-finite proposals are rejected.  One step size for all chains."""
    trace_names = ('is_accepted', 'target_log_prob', 'step_size')

    def __init__(self, target, step_size=0.01, num_leapfrog_steps=5):
        self.target = target
        self.step_size = float(step_size)
        self.num_leapfrog_steps = int(num_leapfrog_steps)

    def bootstrap(self, z):
        lp, g = self.target(z, True)
        return {'target_log_prob': lp, 'grad': g}

    def one_step(self, z, state, generator=None, momenta=None, log_uniform=None):
        eps = self.step_size
        p0 = momenta if momenta is not None else torch.randn(z.shape, generator=generator, device=z.device,
                                                               dtype=z.dtype)
        q, p, g = z, p0, state['grad']
        lp = state['target_log_prob']
        for _ in range(self.num_leapfrog_steps):
            p = p + 0.5 * eps * g
            q = q + eps * p
            lp, g = self.target(q, True)
            p = p + 0.5 * eps * g
        log_ratio = (lp - 0.5 * (p * p).sum(-1)) - (state['target_log_prob'] - 0.5 * (p0 * p0).sum(-1))
        log_ratio = _finite_or_neg_inf(log_ratio)
        if log_uniform is None:
            log_uniform = torch.log(torch.rand(z.shape[0], generator=generator, device=z.device, dtype=z.dtype))
        acc = log_uniform < log_ratio
        z = torch.where(acc[:, None], q, z)
        new_state = {'target_log_prob': torch.where(acc, lp, state['target_log_prob']),
                     'grad': torch.where(acc[:, None], g, state['grad'])}
        trace = {'is_accepted': acc, 'target_log_prob': new_state['target_log_prob'],
                 'step_size': torch.full((), eps, dtype=z.dtype), 'log_accept_ratio': log_ratio}
        return z, new_state, trace


def _ckpt_idxs(n):
    """Checkpoint slots touched by leaf n of an iteratively built binary subtree:
    idx_max = popcount(n >> 1), and the number of completed subtrees ending at an odd
    leaf is the count of trailing one bits."""
    idx_max = bin(n >> 1).count('1')
    trailing = 0
    m = n
    while m & 1:
        trailing += 1
        m >>= 1
    return idx_max - trailing + 1, idx_max


class NoUTurnSamplerKernel:
    """Hoffman & Gelman's No-U-Turn sampler with multinomial trajectory sampling,
    built iteratively (no recursion) and in lock-step over the chain batch: every
    chain takes the same leapfrog schedule and finished chains are masked, so one
    transition costs (max over chains of leapfrogs) fused gradient evaluations."""
    trace_names = ('is_accepted', 'target_log_prob', 'log_accept_ratio', 'step_size',
                   'leapfrogs_taken', 'has_divergence', 'energy')

    def __init__(self, target, step_size=0.01, max_tree_depth=5, max_energy_diff=1000.0,
                 unrolled_leapfrog_steps=1):
        self.target = target
        self.step_size = float(step_size)
        self.max_tree_depth = int(max_tree_depth)
        self.max_energy_diff = float(max_energy_diff)
        self.unrolled = int(unrolled_leapfrog_steps)

    def bootstrap(self, z):
        lp, g = self.target(z, True)
        return {'target_log_prob': lp, 'grad': g}

    @staticmethod
    def _turning(q_left, p_left, q_right, p_right):
        dq = q_right - q_left
        return ((dq * p_left).sum(-1) <= 0) | ((dq * p_right).sum(-1) <= 0)

    def one_step(self, z, state, generator=None):
        C, P = z.shape
        dev, dt = z.device, z.dtype
        eps = self.step_size
        rnd = lambda *s: torch.rand(*s, generator=generator, device=dev, dtype=dt)
        p0 = torch.randn(z.shape, generator=generator, device=dev, dtype=dt)
        lp0, g0 = state['target_log_prob'], state['grad']
        h0 = -lp0 + 0.5 * (p0 * p0).sum(-1)                       # initial energy

        # trajectory ends (left = backward in time, right = forward)
        ql, pl, gl = z.clone(), p0.clone(), g0.clone()
        qr, pr, gr = z.clone(), p0.clone(), g0.clone()
        # current sample and its cached log prob / gradient
        qs, lps, gs = z.clone(), lp0.clone(), g0.clone()
        # log-density minus kinetic energy at the selected point: what tfp's NoUTurnSampler reports as `energy`
        # (compute_hamiltonian of the candidate state; transition_kernels.py:261 traces it)
        es = -h0
        log_w = torch.zeros(C, device=dev, dtype=dt)              # log sum of weights exp(h0 - h)
        active = torch.ones(C, dtype=torch.bool, device=dev)
        diverged = torch.zeros(C, dtype=torch.bool, device=dev)
        sum_acc = torch.zeros(C, device=dev, dtype=dt)
        n_leap = torch.zeros(C, dtype=torch.int64, device=dev)
        moved = torch.zeros(C, dtype=torch.bool, device=dev)
        ckpt_q = torch.zeros(self.max_tree_depth + 1, C, P, device=dev, dtype=dt)
        ckpt_p = torch.zeros_like(ckpt_q)

        for depth in range(self.max_tree_depth):
            if not bool(active.any()):
                break
            go_right = rnd(C) < 0.5
            sgn = torch.where(go_right, 1.0, -1.0).to(dt)[:, None]
            # start the new subtree from the chosen end
            q = torch.where(go_right[:, None], qr, ql)
            p = torch.where(go_right[:, None], pr, pl)
            g = torch.where(go_right[:, None], gr, gl)
            sub_ok = active.clone()                               # chains still building this subtree
            sub_log_w = torch.full((C,), -math.inf, device=dev, dtype=dt)
            sub_q, sub_lp, sub_g, sub_e = qs.clone(), lps.clone(), gs.clone(), es.clone()
            turning_sub = torch.zeros(C, dtype=torch.bool, device=dev)
            for leaf in range(2 ** depth):
                for _ in range(self.unrolled):
                    p = p + 0.5 * eps * sgn * g
                    q = q + eps * sgn * p
                    lp, g = self.target(q, True)
                    p = p + 0.5 * eps * sgn * g
                h = -lp + 0.5 * (p * p).sum(-1)
                dh = h - h0
                dh = torch.where(torch.isfinite(dh), dh, torch.full_like(dh, math.inf))
                div_now = sub_ok & (dh > self.max_energy_diff)
                diverged |= div_now
                ok = sub_ok & ~div_now
                n_leap += sub_ok.to(torch.int64) * self.unrolled
                sum_acc += torch.where(sub_ok, torch.exp(torch.clamp(-dh, max=0.0)), torch.zeros_like(dh))
                # multinomial sample within the subtree (uniform progressive sampling)
                lw = -dh
                new_sub_log_w = torch.logaddexp(sub_log_w, lw)
                take = ok & (torch.log(rnd(C)) < lw - new_sub_log_w)
                sub_q = torch.where(take[:, None], q, sub_q)
                sub_lp = torch.where(take, lp, sub_lp)
                sub_e = torch.where(take, -h, sub_e)
                sub_g = torch.where(take[:, None], g, sub_g)
                sub_log_w = torch.where(ok, new_sub_log_w, sub_log_w)
                # U-turn checks inside the subtree via checkpoints of left ends
                lo, hi = _ckpt_idxs(leaf)
                if leaf % 2 == 0:
                    ckpt_q[hi] = q
                    ckpt_p[hi] = p
                else:
                    for k in range(lo, hi + 1):
                        # ordered in time: for leftward subtrees the checkpoint is the later end
                        a_q = torch.where(go_right[:, None], ckpt_q[k], q)
                        a_p = torch.where(go_right[:, None], ckpt_p[k], p)
                        b_q = torch.where(go_right[:, None], q, ckpt_q[k])
                        b_p = torch.where(go_right[:, None], p, ckpt_p[k])
                        turning_sub |= ok & self._turning(a_q, a_p, b_q, b_p)
                sub_ok = ok & ~turning_sub
                if not bool(sub_ok.any()):
                    break
            # extend the trajectory end for chains that built (part of) the subtree
            ext = active
            qr = torch.where((ext & go_right)[:, None], q, qr)
            pr = torch.where((ext & go_right)[:, None], p, pr)
            gr = torch.where((ext & go_right)[:, None], g, gr)
            ql = torch.where((ext & ~go_right)[:, None], q, ql)
            pl = torch.where((ext & ~go_right)[:, None], p, pl)
            gl = torch.where((ext & ~go_right)[:, None], g, gl)
            # a complete, non-turning, non-divergent subtree may replace the sample
            complete = active & sub_ok
            accept_sub = complete & (torch.log(rnd(C)) < sub_log_w - log_w)     # biased progressive sampling
            qs = torch.where(accept_sub[:, None], sub_q, qs)
            lps = torch.where(accept_sub, sub_lp, lps)
            es = torch.where(accept_sub, sub_e, es)
            gs = torch.where(accept_sub[:, None], sub_g, gs)
            moved |= accept_sub
            log_w = torch.where(complete, torch.logaddexp(log_w, sub_log_w), log_w)
            # stop: incomplete subtree, or U-turn across the whole trajectory
            active = complete & ~self._turning(ql, pl, qr, pr)

        n_safe = torch.clamp(n_leap, min=1).to(dt)
        log_accept_ratio = torch.log(torch.clamp(sum_acc / n_safe, min=1e-30))
        new_state = {'target_log_prob': lps, 'grad': gs}
        trace = {'is_accepted': moved, 'target_log_prob': lps, 'log_accept_ratio': log_accept_ratio,
                 'step_size': torch.full((), eps, dtype=dt), 'leapfrogs_taken': n_leap,
                 'has_divergence': diverged, 'energy': es}
        return qs, new_state, trace


def _log_mean_exp(x):
    return torch.logsumexp(x, 0) - math.log(x.shape[0])


def _adapting_many_steps(wrapper, z, state, n, generator=None):
    """`many_steps` of a step-size adaptation wrapper: transitions that still adapt go one by one
    (the rule needs every transition's acceptance probability on the host), the rest in one call
    of the inner kernel."""
    zs_parts, trace_parts = [], []
    done = 0
    while done < n and wrapper.step < wrapper.num_adaptation_steps:
        z, state, tr = wrapper.one_step(z, state, generator=generator)
        zs_parts.append(z[None])
        trace_parts.append({k: torch.as_tensor(v)[None] for k, v in tr.items()})
        done += 1
    if done < n:
        zs, z, state, tr = wrapper.inner.many_steps(z, state, n - done, generator=generator)
        wrapper.step += n - done
        zs_parts.append(zs)
        trace_parts.append(tr)
    keys = trace_parts[0].keys()
    traces = {k: torch.cat([p[k].to(z.device) for p in trace_parts], 0) for k in keys}
    return torch.cat(zs_parts, 0), z, state, traces


class SimpleStepSizeAdaptation:
    """Multiplicative step-size rule of tfp.mcmc.SimpleStepSizeAdaptation: while adapting,
    step *= (1 + rate) when the chain-averaged acceptance probability exceeds the target,
    else step /= (1 + rate)."""

    def __init__(self, inner, num_adaptation_steps, target_accept_prob=0.75, adaptation_rate=0.01):
        self.inner = inner
        self.num_adaptation_steps = int(num_adaptation_steps)
        self.target = float(target_accept_prob)
        self.rate = float(adaptation_rate)
        self.step = 0
        self.trace_names = inner.trace_names
        self.supports_many_steps = getattr(inner, 'supports_many_steps', False)

    def bootstrap(self, z):
        return self.inner.bootstrap(z)

    def many_steps(self, z, state, n, generator=None):
        return _adapting_many_steps(self, z, state, n, generator)

    def one_step(self, z, state, generator=None):
        z, state, trace = self.inner.one_step(z, state, generator=generator)
        if self.step < self.num_adaptation_steps:
            lap = _log_mean_exp(torch.clamp(trace['log_accept_ratio'], max=0.0))
            adj = self.rate if float(lap) > math.log(self.target) else -self.rate / (1.0 + self.rate)
            self.inner.step_size *= (1.0 + adj)
        self.step += 1
        return z, state, trace


class DualAveragingStepSizeAdaptation:
    """Nesterov dual averaging of log step size (Hoffman & Gelman 2014, section 3.2) with
    the constants of bayes_vi/inference/mcmc/stepsize_adaptation_kernels.py:72-76:
    exploration_shrinkage gamma = 0.05, step_count_smoothing t0 = 10, decay_rate kappa = 0.75,
    shrinkage_target mu = log(10 * initial step size) unless given.  The acceptance
    probability is averaged over all chains, so the chains share one step size; after
    num_adaptation_steps the averaged iterate is frozen in."""

    def __init__(self, inner, num_adaptation_steps, target_accept_prob=0.75, exploration_shrinkage=0.05,
                 shrinkage_target=None, step_count_smoothing=10, decay_rate=0.75):
        self.inner = inner
        self.num_adaptation_steps = int(num_adaptation_steps)
        self.target = float(target_accept_prob)
        self.gamma = float(exploration_shrinkage)
        self.t0 = float(step_count_smoothing)
        self.kappa = float(decay_rate)
        self.mu = math.log(10.0 * inner.step_size) if shrinkage_target is None else float(shrinkage_target)
        self.error_sum = 0.0
        self.log_averaging_step = 0.0
        self.step = 0
        self.trace_names = inner.trace_names
        self.supports_many_steps = getattr(inner, 'supports_many_steps', False)

    def bootstrap(self, z):
        return self.inner.bootstrap(z)

    def many_steps(self, z, state, n, generator=None):
        return _adapting_many_steps(self, z, state, n, generator)

    def one_step(self, z, state, generator=None):
        z, state, trace = self.inner.one_step(z, state, generator=generator)
        if self.step < self.num_adaptation_steps:
            accept = float(torch.exp(_log_mean_exp(torch.clamp(trace['log_accept_ratio'], max=0.0))))
            t = self.step + 1
            self.error_sum += self.target - accept
            log_step = self.mu - math.sqrt(t) / self.gamma * self.error_sum / (t + self.t0)
            eta = t ** (-self.kappa)
            self.log_averaging_step = eta * log_step + (1.0 - eta) * self.log_averaging_step
            self.inner.step_size = math.exp(log_step)
            if t == self.num_adaptation_steps:
                self.inner.step_size = math.exp(self.log_averaging_step)
        self.step += 1
        return z, state, trace


class KernelResults(dict):
    """What a kernel reports about its transitions (tfp's "kernel results"): fields by name, also readable as
    attributes; a wrapping kernel's results hold the wrapped kernel's under `inner_results`, which is what the
    reference's unwrapping trace functions walk down (transition_kernels.py:30, 37; mcmc.py:157)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)


def kernel_results(kernel, traces):
    """The results of `kernel` for the transitions whose traced fields `traces` holds, nested the way the kernel
    objects are nested (step-size adaptation around the sampling kernel)."""
    if hasattr(kernel, 'inner'):
        return KernelResults(inner_results=kernel_results(kernel.inner, traces),
                             new_step_size=getattr(kernel.inner, 'step_size', None))
    return KernelResults(traces)


def sample_chain(kernel, z0, num_results, num_burnin_steps, generator=None, progress=None, trace_fn=None):
    """tfp.mcmc.sample_chain restated: burn in, then collect `num_results` states and the
    kernel's traced quantities.  Returns (samples[num_results, C, P], {name: [num_results, ...]}); with
    `trace_fn(states, kernel_results)` (mcmc.py:167-173) the second entry is what it returns for the collected
    transitions (every field stacked over them)."""
    if trace_fn is not None:
        samples, traces = sample_chain(kernel, z0, num_results, num_burnin_steps, generator=generator,
                                       progress=progress)
        return samples, trace_fn(samples, kernel_results(kernel, traces))
    z = z0
    state = kernel.bootstrap(z)
    if not bool(torch.isfinite(state['target_log_prob']).all()):
        raise ValueError('initial state has non-finite log probability: %s' % state['target_log_prob'])
    total = num_burnin_steps + num_results
    if getattr(kernel, 'supports_many_steps', False):
        # kernels that run whole stretches of transitions per launch: burn-in, then the results
        if num_burnin_steps > 0:
            _, z, state, _ = kernel.many_steps(z, state, num_burnin_steps, generator=generator)
            if progress is not None:
                progress(num_burnin_steps, total)
        samples, z, state, tr = kernel.many_steps(z, state, num_results, generator=generator)
        if progress is not None:
            progress(total, total)
        return samples, {name: tr[name] for name in kernel.trace_names}
    samples = torch.empty((num_results,) + tuple(z.shape), dtype=z.dtype, device=z.device)
    traces = {}
    for i in range(total):
        z, state, tr = kernel.one_step(z, state, generator=generator)
        if i >= num_burnin_steps:
            j = i - num_burnin_steps
            samples[j] = z
            for name in kernel.trace_names:
                v = tr[name]
                if name not in traces:
                    traces[name] = torch.empty((num_results,) + tuple(v.shape), dtype=v.dtype, device=v.device)
                traces[name][j] = v
        if progress is not None:
            progress(i + 1, total)
    return samples, traces
