"""Surrogate posteriors (mirror of bayes_vi/inference/vi/surrogate_posteriors.py).

A surrogate lives in CONSTRAINED parameter space: a base distribution in unconstrained space
pushed through the model's `blockwise_constraining_bijector` (surrogate_posteriors.py:120-123,
131-134, 203-205).  `sample_and_log_prob(n)` draws reparameterised samples and their log
density in one go (what tfp.vi.monte_carlo_variational_loss needs).
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

LN2PI = math.log(2.0 * math.pi)


def _softplus_inverse(x):
    return x + torch.log(-torch.expm1(-x))


def _modules_reachable_from(fn):
    """nn.Modules a callable is, or closes over (one level of closure cells and default arguments)."""
    if fn is None:
        return []
    if isinstance(fn, nn.Module):
        return [fn]
    found = []
    cells = [c.cell_contents for c in (getattr(fn, '__closure__', None) or ()) if c is not None]
    for obj in cells + list(getattr(fn, '__defaults__', None) or ()):
        if isinstance(obj, nn.Module) and not any(obj is f for f in found):
            found.append(obj)
    return found


class SurrogatePosterior(nn.Module):
    """Base class (surrogate_posteriors.py:10-94)."""

    def __init__(self, model):
        super().__init__()
        self.model = model
        self.unconstrained_event_ndims = self.model.flat_unconstrained_param_event_ndims
        self.event_ndims = self.model.flat_param_event_ndims
        dtypes = list(self.model.dtypes.values())
        if len(set(dtypes)) == 1:
            self.dtype = dtypes[0]
        else:
            raise ValueError('Model has incompatible dtypes: {}'.format(set(dtypes)))
        self.extra_ndims = 0

    def base_noise(self, n, dim):
        """Standard-normal draws [n, dim] that the reparameterised sampler transforms.  Tests replace this attribute
        to evaluate a step on injected noise."""
        dev = next((p.device for p in self.parameters()), torch.device('cpu'))
        return torch.randn(n, dim, device=dev, dtype=torch.float32)

    # reshape_sample_bijector.forward: flat constrained sample -> list of named parts
    def reshape_sample(self, q):
        return self.model.split_state(q)

    def sample_unconstrained(self, n):
        """(u[n, Pu + extra], log q_u(u)[n]) -- implemented by subclasses."""
        raise NotImplementedError

    def sample_and_log_prob(self, n):
        """Constrained samples [n, P + extra] and their log density under the surrogate."""
        u, logq = self.sample_unconstrained(n)
        bij = self.model.blockwise_constraining_bijector
        pu = self.unconstrained_event_ndims
        z, a = u[:, :pu], u[:, pu:]
        theta = bij.forward(z)
        logq = logq - bij.forward_log_det_jacobian(z)
        return torch.cat([theta, a], -1), logq

    def sample(self, n):
        with torch.no_grad():
            return self.sample_and_log_prob(n)[0]

    def approx_joint_marginal_posteriors(self, num_samples_to_approx_marginals):
        """Empirical marginals of each parameter from surrogate samples (surrogate_posteriors.py:51-70)."""
        q = self.sample(num_samples_to_approx_marginals)[:, :self.event_ndims]
        return self.model.get_param_distributions(param_samples=self.reshape_sample(q))

    def get_corrected_target_log_prob_fn(self, target_log_prob_fn):
        """For augmented flows the target gains the lift distribution's log density of the extra
        dimensions given the unconstrained parameters (surrogate_posteriors.py:72-94)."""
        if self.extra_ndims > 0:
            def corrected(sample):
                q, a = sample[:, :self.event_ndims], sample[:, self.event_ndims:]
                z = self.model.blockwise_constraining_bijector.inverse(q)
                return target_log_prob_fn(q) + self.posterior_lift_log_prob(z, a)
            return corrected
        return target_log_prob_fn


class ADVI(SurrogatePosterior):
    """Automatic Differentiation VI (surrogate_posteriors.py:97-134): Gaussian in unconstrained
    space, mean-field (softplus-parameterised diagonal, initial scale 1) or full rank (scale_tril
    initialised to the identity, softplus diagonal + 1e-5), loc ~ N(0, 1)."""

    def __init__(self, model, mean_field=False, generator=None):
        super().__init__(model=model)
        pu = self.unconstrained_event_ndims
        self.mean_field = mean_field
        self.loc = nn.Parameter(torch.randn(pu, generator=generator))
        if mean_field:
            self.raw_scale = nn.Parameter(_softplus_inverse(torch.ones(pu)))
        else:
            raw = torch.zeros(pu, pu)
            raw.diagonal().fill_(float(_softplus_inverse(torch.tensor(1.0 - 1e-5))))
            self.raw_tril = nn.Parameter(raw)

    def scale_tril(self):
        L = torch.tril(self.raw_tril, -1)
        return L + torch.diag(F.softplus(self.raw_tril.diagonal()) + 1e-5)

    def sample_unconstrained(self, n):
        pu = self.unconstrained_event_ndims
        eps = self.base_noise(n, pu)
        if self.mean_field:
            sc = F.softplus(self.raw_scale)
            z = self.loc + sc * eps
            logq = (-0.5 * eps * eps - 0.5 * LN2PI - torch.log(sc)).sum(-1)
        else:
            L = self.scale_tril()
            z = self.loc + eps @ L.T
            logq = (-0.5 * eps * eps - 0.5 * LN2PI).sum(-1) - torch.log(L.diagonal()).sum()
        return z, logq


class NormalizingFlow(SurrogatePosterior):
    """Normalizing-flow surrogate, optionally augmented with `extra_ndims` auxiliary dimensions
    (surrogate_posteriors.py:137-212): base N(0, I) on R^{Pu + extra} -> flow bijector -> constrain
    the first Pu coordinates.  `posterior_lift_distribution(z)` defaults to N(0, I) on the extras."""

    def __init__(self, model, flow_bijector, base_distribution=None, extra_ndims=None,
                 posterior_lift_distribution=None):
        super().__init__(model=model)
        self.flow_bijector = flow_bijector
        if extra_ndims and extra_ndims < 0:
            raise ValueError('`extra_dims` have to be `None` or  `>=0`.')
        self.extra_ndims = extra_ndims if extra_ndims else 0
        self.base_distribution = base_distribution       # object with sample_and_log_prob(n) or None
        self.posterior_lift_distribution = posterior_lift_distribution
        # the reference trains every variable the loss touches (tape.watched_variables(), vi.py:69), which includes
        # networks a lift callable closes over (linear-regression-example.ipynb cell 13): register them here so that
        # `parameters()` finds them
        self.lift_networks = nn.ModuleList(_modules_reachable_from(posterior_lift_distribution))

    def posterior_lift_log_prob(self, z, a):
        if self.posterior_lift_distribution is not None:
            return self.posterior_lift_distribution(z).log_prob(a)
        return (-0.5 * a * a - 0.5 * LN2PI).sum(-1)

    def sample_unconstrained(self, n):
        dim = self.unconstrained_event_ndims + self.extra_ndims
        if self.base_distribution is not None:
            x, logp = self.base_distribution.sample_and_log_prob(n)
        else:
            x = self.base_noise(n, dim)
            logp = (-0.5 * x * x - 0.5 * LN2PI).sum(-1)
        u = self.flow_bijector.forward(x)
        return u, logp - self.flow_bijector.forward_log_det_jacobian(x)


class TrainableNormalDiag(nn.Module):
    """A trainable diagonal-Gaussian base distribution for flows
    (README.md:668: MultivariateNormalDiag(loc=Variable, scale_diag=Variable))."""

    def __init__(self, loc, scale_diag):
        super().__init__()
        self.loc = nn.Parameter(torch.as_tensor(loc, dtype=torch.float32).clone())
        self.scale_diag = nn.Parameter(torch.as_tensor(scale_diag, dtype=torch.float32).clone())

    def sample_and_log_prob(self, n):
        eps = torch.randn(n, self.loc.shape[0], device=self.loc.device)
        x = self.loc + self.scale_diag * eps
        return x, (-0.5 * eps * eps - 0.5 * LN2PI - torch.log(self.scale_diag.abs())).sum(-1)
