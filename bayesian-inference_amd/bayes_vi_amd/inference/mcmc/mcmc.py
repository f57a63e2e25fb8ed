"""`MCMC`: posterior sampling (mirror of bayes_vi/inference/mcmc/mcmc.py)."""
import collections
import os

import torch
import torch.distributed as dist

from ...fused_target import replicate_from_rank0
from ...utils import compose
from ..inference import Inference
from .sample_results import SampleResult
from .samplers import sample_chain
from .transition_kernels import RandomWalkMetropolis


class MCMC(Inference):
    """Same construction and `fit` signature as the reference (mcmc.py:39, 101-102).

    The whole dataset is taken as one batch and placed on the GPU once (mcmc.py:52); the
    joint is conditioned on the features (mcmc.py:55); `target_log_prob` is the closure of
    mcmc.py:57-62, here backed by the fused kernels.  Chains run in unconstrained space
    through the model's constraining bijectors, as TransformedTransitionKernel does
    (mcmc.py:153-156).  `group` (optional torch.distributed process group) marks this
    process's dataset as one row shard of a larger one: partial sums are all-reduced.
    """

    def __init__(self, model, dataset, transition_kernel=None, group=None, precision=None):
        super().__init__(model=model, dataset=dataset)
        self.features, self.targets = dataset.full_batch()
        self.distribution = self.model.get_joint_distribution(num_samples=self.targets.shape[0],
                                                               features=self.features, group=group,
                                                               precision=precision)
        self.posterior = self.distribution.posterior(self.targets)
        self.device = self.posterior.likelihood.device
        self.group = group
        self.transition_kernel = transition_kernel if transition_kernel is not None else RandomWalkMetropolis()

    def target_log_prob(self, state):
        """Unnormalised log posterior of a constrained parameter sample given as a list of parts."""
        theta = self.model.merge_state_parts([torch.as_tensor(s) for s in state]).to(self.device)
        return self.posterior.log_prob(theta.to(torch.float32))

    def _process_initial_state(self, initial_state, num_chains, generator=None):
        """None: U(-1, 1) in unconstrained space, then constrained (mcmc.py:66-81)."""
        if initial_state is None:
            z = torch.rand(num_chains, self.model.flat_unconstrained_param_event_ndims, generator=generator) * 2 - 1
            return z.to(self.device)
        if isinstance(initial_state, (list, collections.OrderedDict)):
            parts = initial_state if isinstance(initial_state, list) else list(initial_state.values())
            theta = self.model.merge_state_parts(parts).to(self.device).to(torch.float32)
            return self.model.blockwise_constraining_bijector.inverse(theta)
        raise ValueError('`initial_state` has to be in [None, list, collections.OrderedDict]; '
                         'was {}'.format(initial_state))

    def fit(self, initial_state=None, num_chains=4, num_samples=4000, num_burnin_steps=1000,
            merge_state_parts=False, progress_bar=True, seed=None):
        """Returns a `SampleResult`: samples as a list of [num_samples, num_chains, *event] per
        parameter (or one merged [num_samples, num_chains, P] entry with merge_state_parts) and the
        kernel's traced metrics (mcmc.py:101-185)."""
        # One seed drives everything random in the run: the initial state, the counter-based device streams and
        # the generators of the host kernels.  Without `seed` a fresh one is drawn, so that repeated fits are
        # independent runs (as the reference's are); with rows sharded over ranks it is rank 0's, so that the
        # replicated chains see the same momenta and accept draws on every rank whatever kernel is selected.
        if seed is None:
            seed = int.from_bytes(os.urandom(7), 'little')
        wire = torch.tensor([int(seed)], dtype=torch.int64)
        if self.group is not None and dist.is_initialized() and dist.get_backend(self.group) != 'gloo':
            wire = wire.to(self.device)
        seed = int(replicate_from_rank0(wire, self.group)[0])
        gen = torch.Generator(device='cpu'); gen.manual_seed(seed)
        z0 = self._process_initial_state(initial_state, num_chains, generator=gen)
        z0 = replicate_from_rank0(z0.contiguous(), self.group)
        dgen = torch.Generator(device=self.device); dgen.manual_seed((seed + 1) % (1 << 63))

        target = self.posterior.unconstrained_log_prob_and_grad
        kernel, trace_fns, trace_metrics = self.transition_kernel(target)

        progress = None
        if progress_bar:
            try:
                from tqdm import tqdm
                bar = tqdm(total=num_samples + num_burnin_steps, leave=False)
                progress = lambda i, n: bar.update(i - bar.n)
            except Exception:
                progress = None
        zs, trace = sample_chain(kernel, z0, num_samples, num_burnin_steps, generator=dgen, progress=progress,
                                 trace_fn=compose(list(reversed(trace_fns))))
        if progress_bar and progress is not None:
            bar.close()

        theta = self.model.blockwise_constraining_bijector.forward(zs)       # [S, C, P] constrained
        samples = [theta] if merge_state_parts else self.model.split_state(theta)
        if merge_state_parts:
            samples = self.model.split_state(theta)   # the reference un-merges before returning (mcmc.py:176-179)
        trace_dict = {name: values for name, values in zip(trace_metrics, trace)}
        return SampleResult(samples, trace_dict)
