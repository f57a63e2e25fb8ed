"""Step-size adaptation configuration (mirror of
bayes_vi/inference/mcmc/stepsize_adaptation_kernels.py): same class names, arguments and
defaults; `__call__(inner_kernel)` wraps a kernel of `samplers` instead of a TFP kernel."""
from . import samplers


class StepSizeAdaptationKernel:
    """Base class for step size adaptation kernels."""

    def __init__(self, num_adaptation_steps, target_accept_prob, step_size_setter_fn=None,
                 step_size_getter_fn=None, log_accept_prob_getter_fn=None):
        self.num_adaptation_steps = num_adaptation_steps
        self.target_accept_prob = target_accept_prob
        self.step_size_setter_fn = step_size_setter_fn
        self.step_size_getter_fn = step_size_getter_fn
        self.log_accept_prob_getter_fn = log_accept_prob_getter_fn

    def __str__(self):
        return '%s(%s)' % (type(self).__name__, ', '.join('%s=%s' % item for item in vars(self).items()))

    def __call__(self, *args, **kwargs):
        raise NotImplementedError('Not yet Implemented')


class SimpleStepSizeAdaptation(StepSizeAdaptationKernel):
    """Multiplicative adaptation (stepsize_adaptation_kernels.py:31-61)."""

    def __init__(self, num_adaptation_steps, target_accept_prob=0.75, adaptation_rate=0.01,
                 step_size_setter_fn=None, step_size_getter_fn=None, log_accept_prob_getter_fn=None,
                 validate_args=False, name=None):
        super().__init__(num_adaptation_steps, target_accept_prob, step_size_setter_fn, step_size_getter_fn,
                         log_accept_prob_getter_fn)
        self.adaptation_rate = adaptation_rate
        self.name = name

    def __call__(self, inner_kernel):
        return samplers.SimpleStepSizeAdaptation(inner_kernel, self.num_adaptation_steps,
                                                 target_accept_prob=self.target_accept_prob,
                                                 adaptation_rate=self.adaptation_rate)


class DualAveragingStepSizeAdaptation(StepSizeAdaptationKernel):
    """Dual averaging (stepsize_adaptation_kernels.py:64-104; defaults :72-76)."""

    def __init__(self, num_adaptation_steps, target_accept_prob=0.75, exploration_shrinkage=0.05,
                 shrinkage_target=None, step_count_smoothing=10, decay_rate=0.75,
                 step_size_setter_fn=None, step_size_getter_fn=None, log_accept_prob_getter_fn=None,
                 validate_args=False, name=None):
        super().__init__(num_adaptation_steps, target_accept_prob, step_size_setter_fn, step_size_getter_fn,
                         log_accept_prob_getter_fn)
        self.exploration_shrinkage = exploration_shrinkage
        self.shrinkage_target = shrinkage_target
        self.step_count_smoothing = step_count_smoothing
        self.decay_rate = decay_rate
        self.name = name

    def __call__(self, inner_kernel):
        return samplers.DualAveragingStepSizeAdaptation(
            inner_kernel, self.num_adaptation_steps, target_accept_prob=self.target_accept_prob,
            exploration_shrinkage=self.exploration_shrinkage, shrinkage_target=self.shrinkage_target,
            step_count_smoothing=self.step_count_smoothing, decay_rate=self.decay_rate)
