"""Markov transition kernel configuration (mirror of
bayes_vi/inference/mcmc/transition_kernels.py): same class names, constructor arguments,
defaults and traced metrics.  Calling a kernel object with a target builds the batched
kernel of `samplers` (the reference builds a functools.partial over tfp.mcmc.*)."""
from . import samplers
from .stepsize_adaptation_kernels import StepSizeAdaptationKernel


class TransitionKernel:
    """Base class for Markov transition kernels (transition_kernels.py:12-38)."""

    def __init__(self, name=None):
        self.name = name

    def __str__(self):
        return '%s(%s)' % (type(self).__name__, ', '.join('%s=%s' % item for item in vars(self).items()))

    def _make(self, target):
        raise NotImplementedError

    @classmethod
    def trace_fn(cls, _, pkr):
        """The traced fields of this kernel's results, in `trace_fn_metrics` order
        (transition_kernels.py:131-137, 252-262, 324-326)."""
        return tuple(pkr[name] for name in cls.TRACED)

    def __call__(self, target_log_prob_fn):
        """target_log_prob_fn(z, want_grad) -> (log_prob[C], grad[C, P] or None) in unconstrained space.
        Returns (kernel, trace_fns, trace_fn_metrics) as the reference does (transition_kernels.py:24-38):
        `trace_fns` starts with this kernel's own trace function and gains one unwrapping function per kernel
        wrapped around it; composed outermost first (`compose(list(reversed(trace_fns)))`, mcmc.py:172) they map
        (states, kernel_results) to the tuple of traced values."""
        if getattr(self, 'transforming_bijector', None) is not None:
            raise NotImplementedError('transforming_bijector (trainable preconditioning) is not part of the '
                                      'fused hot path')
        kernel = self._make(target_log_prob_fn)
        trace_fns = [self.trace_fn]
        ada = getattr(self, 'stepsize_adaptation_kernel', None)
        if isinstance(ada, StepSizeAdaptationKernel):
            kernel = ada(kernel)
            trace_fns.append(lambda _, pkr: (_, pkr.inner_results))
        return kernel, trace_fns, list(self.trace_fn_metrics)


class HamiltonianMonteCarlo(TransitionKernel):
    """HMC (transition_kernels.py:41-137; defaults step_size=0.01, num_leapfrog_steps=5)."""
    TRACED = ('is_accepted', 'target_log_prob', 'step_size')

    def __init__(self, step_size=0.01, num_leapfrog_steps=5, transforming_bijector=None,
                 stepsize_adaptation_kernel=None, state_gradients_are_stopped=False, name=None):
        super().__init__(name)
        self.transforming_bijector = transforming_bijector
        self.stepsize_adaptation_kernel = stepsize_adaptation_kernel
        self.step_size = step_size
        self.num_leapfrog_steps = num_leapfrog_steps
        self.state_gradients_are_stopped = state_gradients_are_stopped
        self.trace_fn_metrics = list(self.TRACED)

    def _make(self, target):
        # a target that is a GPU `Posterior` whose priors/bijectors the device kernels implement
        # runs the whole transition on the stream; anything else takes the generic torch kernel
        post = getattr(target, '__self__', None)
        if post is not None and hasattr(post, 'device_blocks') and post.likelihood.device.type == 'cuda' \
                and post.device_blocks() is not None:
            from .device_hmc import (DeviceHamiltonianMonteCarloKernel, FusedHamiltonianMonteCarloKernel,
                                     fused_family)
            fam = fused_family(post)
            if fam is not None:     # small dataset: whole transitions in one launch
                return FusedHamiltonianMonteCarloKernel(post, fam, self.step_size, self.num_leapfrog_steps)
            return DeviceHamiltonianMonteCarloKernel(post, self.step_size, self.num_leapfrog_steps)
        return samplers.HamiltonianMonteCarloKernel(target, self.step_size, self.num_leapfrog_steps)


class NoUTurnSampler(TransitionKernel):
    """NUTS (transition_kernels.py:140-262; defaults step_size=0.01, max_tree_depth=5,
    max_energy_diff=1000, unrolled_leapfrog_steps=1)."""
    TRACED = ('is_accepted', 'target_log_prob', 'log_accept_ratio', 'step_size', 'leapfrogs_taken',
              'has_divergence', 'energy')

    def __init__(self, step_size=0.01, max_tree_depth=5, max_energy_diff=1000.0, unrolled_leapfrog_steps=1,
                 parallel_iterations=10, transforming_bijector=None, stepsize_adaptation_kernel=None, name=None):
        super().__init__(name)
        self.transforming_bijector = transforming_bijector
        self.stepsize_adaptation_kernel = stepsize_adaptation_kernel
        self.step_size = step_size
        self.max_tree_depth = max_tree_depth
        self.max_energy_diff = max_energy_diff
        self.unrolled_leapfrog_steps = unrolled_leapfrog_steps
        self.parallel_iterations = parallel_iterations
        self.trace_fn_metrics = list(self.TRACED)

    def _make(self, target):
        # small dataset on the GPU with built-in priors / bijectors: whole transitions in one launch
        post = getattr(target, '__self__', None)
        if post is not None and hasattr(post, 'device_blocks') and post.likelihood.device.type == 'cuda' \
                and self.unrolled_leapfrog_steps == 1 and self.max_tree_depth <= 10:
            from .device_hmc import BatchedDeviceNoUTurnSamplerKernel, FusedNoUTurnSamplerKernel, fused_family
            fam = fused_family(post)
            if fam is not None:
                return FusedNoUTurnSamplerKernel(post, fam, self.step_size, self.max_tree_depth,
                                                 self.max_energy_diff)
            if post.device_blocks() is not None:    # any size: tree bookkeeping on the device, batched data passes
                return BatchedDeviceNoUTurnSamplerKernel(post, self.step_size, self.max_tree_depth,
                                                         self.max_energy_diff)
        return samplers.NoUTurnSamplerKernel(target, self.step_size, self.max_tree_depth, self.max_energy_diff,
                                             self.unrolled_leapfrog_steps)


class RandomWalkMetropolis(TransitionKernel):
    """Random-walk Metropolis (transition_kernels.py:265-326); default proposal N(0, 1) per coordinate."""
    TRACED = ('is_accepted',)

    def __init__(self, transforming_bijector=None, new_state_fn=None, name=None):
        super().__init__(name)
        self.transforming_bijector = transforming_bijector
        self.new_state_fn = new_state_fn
        self.trace_fn_metrics = list(self.TRACED)

    def _make(self, target):
        post = getattr(target, '__self__', None)
        if self.new_state_fn is None and post is not None and hasattr(post, 'device_blocks') \
                and post.likelihood.device.type == 'cuda':
            from .device_hmc import (DeviceRandomWalkMetropolisKernel, FusedRandomWalkMetropolisKernel,
                                     fused_family)
            fam = fused_family(post)
            if fam is not None:     # small dataset, default proposal: whole runs of transitions per launch
                return FusedRandomWalkMetropolisKernel(post, fam)
            if post.device_blocks() is not None:    # any size: propose / accept on the stream, no host sync
                return DeviceRandomWalkMetropolisKernel(post)
        return samplers.RandomWalkMetropolisKernel(target, new_state_fn=self.new_state_fn)
