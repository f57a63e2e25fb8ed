"""A small distribution vocabulary with TFP-shaped constructors.

The reference builds its models from `tfp.distributions` objects
(README.md:114-120, 747-756; exploratory_experiments/*.ipynb); this module
offers the same names and keyword arguments for the ones its models use, on
torch tensors.  Priors are evaluated here (O(#parameters) work per chain);
likelihoods are only *described* by these objects -- `Model` recognises the
description and routes the sum over datapoints to the fused HIP kernels.

Parameters may be python floats, torch tensors, or the symbolic expressions of
`bayes_vi_amd.symbolic` (which is how a likelihood callable is recognised
without evaluating it on data).
"""
from __future__ import annotations

import math

import torch

from . import bijectors as tfb

LN2PI = math.log(2.0 * math.pi)


def _t(v, like=None):
    if isinstance(v, torch.Tensor):
        return v
    try:
        return torch.as_tensor(v, dtype=torch.float32)
    except (TypeError, ValueError, RuntimeError):
        return v  # symbolic expression


_DEVICE_COPIES = {}


def _on(t, device):
    """Hyper-parameter tensor `t` on `device`, copied once and cached (a per-call `.to(device)` of a host
    tensor is a blocking upload: slow in a sampling loop and not allowed under HIP graph capture)."""
    if t.device == device:
        return t
    key = (id(t), str(device))
    hit = _DEVICE_COPIES.get(key)
    if hit is None or hit[0] is not t:
        hit = (t, t.to(device))
        _DEVICE_COPIES[key] = hit
    return hit[1]


def _is_concrete(*vs):
    return all(isinstance(v, torch.Tensor) for v in vs)


class Distribution:
    """Common surface: batch_shape/event_shape, dtype, log_prob, sample,
    default event-space bijector, and the closed-form score used by samplers."""
    dtype = torch.float32

    @property
    def batch_shape(self):
        return torch.Size(())

    @property
    def event_shape(self):
        return torch.Size(())

    def param_event_shape(self):
        """Shape of one draw = batch_shape + event_shape (what AutoBatched treats as the event)."""
        return torch.Size(tuple(self.batch_shape) + tuple(self.event_shape))

    def log_prob(self, x):
        raise NotImplementedError

    def log_prob_flat(self, x):
        """x[..., n] with n = prod(param_event_shape): summed log density [...]."""
        raise NotImplementedError

    def score_flat(self, x):
        """d log_prob_flat / d x, closed form."""
        raise NotImplementedError

    def sample(self, sample_shape=(), generator=None):
        raise NotImplementedError

    def default_bijector(self):
        return tfb.Identity()


class Normal(Distribution):
    def __init__(self, loc, scale, name='Normal'):
        self.loc, self.scale, self.name = _t(loc), _t(scale), name

    @property
    def batch_shape(self):
        if not _is_concrete(self.loc, self.scale):
            return torch.Size(())
        return torch.broadcast_shapes(self.loc.shape, self.scale.shape)

    def log_prob(self, x):
        z = (x - _on(self.loc, x.device)) / _on(self.scale, x.device)
        return -0.5 * z * z - 0.5 * LN2PI - torch.log(_on(self.scale, x.device))

    def log_prob_flat(self, x):
        loc = _on(self.loc, x.device).expand(self.batch_shape).reshape(-1) if self.loc.dim() else _on(self.loc, x.device)
        sc = _on(self.scale, x.device).expand(self.batch_shape).reshape(-1) if self.scale.dim() else _on(self.scale, x.device)
        z = (x - loc) / sc
        return (-0.5 * z * z - 0.5 * LN2PI - torch.log(sc)).sum(-1)

    def score_flat(self, x):
        loc = _on(self.loc, x.device).expand(self.batch_shape).reshape(-1) if self.loc.dim() else _on(self.loc, x.device)
        sc = _on(self.scale, x.device).expand(self.batch_shape).reshape(-1) if self.scale.dim() else _on(self.scale, x.device)
        return -(x - loc) / (sc * sc)

    def sample(self, sample_shape=(), generator=None):
        shape = tuple(sample_shape) + tuple(self.batch_shape)
        return self.loc + self.scale * torch.randn(shape, generator=generator)


class HalfNormal(Distribution):
    def __init__(self, scale, name='HalfNormal'):
        self.scale, self.name = _t(scale), name

    @property
    def batch_shape(self):
        return self.scale.shape if _is_concrete(self.scale) else torch.Size(())

    def _sc(self, x):
        sc = _on(self.scale, x.device)
        return sc.reshape(-1) if sc.dim() else sc

    def log_prob(self, x):
        sc = _on(self.scale, x.device)
        lp = 0.5 * math.log(2.0 / math.pi) - torch.log(sc) - 0.5 * (x / sc) ** 2
        return torch.where(x < 0, torch.full_like(lp, -math.inf), lp)

    def log_prob_flat(self, x):
        sc = self._sc(x)
        lp = 0.5 * math.log(2.0 / math.pi) - torch.log(sc) - 0.5 * (x / sc) ** 2
        return torch.where(x < 0, torch.full_like(lp, -math.inf), lp).sum(-1)

    def score_flat(self, x):
        sc = self._sc(x)
        return -x / (sc * sc)

    def sample(self, sample_shape=(), generator=None):
        shape = tuple(sample_shape) + tuple(self.batch_shape)
        return (self.scale * torch.randn(shape, generator=generator)).abs()

    def default_bijector(self):
        return tfb.Softplus()


class Dirichlet(Distribution):
    def __init__(self, concentration, name='Dirichlet'):
        self.concentration, self.name = _t(concentration), name

    @property
    def event_shape(self):
        return self.concentration.shape[-1:]

    @property
    def batch_shape(self):
        return self.concentration.shape[:-1]

    def log_prob(self, x):
        a = _on(self.concentration, x.device)
        norm = torch.lgamma(a.sum(-1)) - torch.lgamma(a).sum(-1)
        return norm + ((a - 1.0) * torch.log(x)).sum(-1)

    def log_prob_flat(self, x):
        return self.log_prob(x)

    def score_flat(self, x):
        return (_on(self.concentration, x.device) - 1.0) / x

    def sample(self, sample_shape=(), generator=None):
        a = self.concentration.expand(tuple(sample_shape) + tuple(self.concentration.shape))
        g = torch._standard_gamma(a) if generator is None else torch._standard_gamma(a)
        return g / g.sum(-1, keepdim=True)

    def default_bijector(self):
        return tfb.SoftmaxCentered()


class Categorical(Distribution):
    """Only used to describe mixture weights: Categorical(probs=weights)."""

    def __init__(self, probs=None, logits=None, name='Categorical'):
        if (probs is None) == (logits is None):
            raise ValueError('Must pass probs or logits, but not both.')
        self.probs, self.logits, self.name = probs, logits, name


class MultivariateNormalTriL(Distribution):
    def __init__(self, loc=None, scale_tril=None, name='MultivariateNormalTriL'):
        self.loc, self.scale_tril, self.name = _t(loc), _t(scale_tril), name

    def sample(self, sample_shape=(), generator=None):
        """loc[..., D] + scale_tril[..., D, D] eps, eps ~ N(0, I): concrete parameters only."""
        shape = tuple(sample_shape) + tuple(self.loc.shape)
        eps = torch.randn(shape, generator=generator).to(self.loc.device)
        return self.loc + (torch.tril(self.scale_tril) @ eps[..., None])[..., 0]


class MultivariateNormalDiag(Distribution):
    """Independent normals over the last axis, tfd.MultivariateNormalDiag(loc, scale_diag): what the reference's
    experiments return from `posterior_lift_distribution` (linear-regression-example.ipynb cell 13) and use as a
    trainable flow base (README.md:668).  Differentiable in loc and scale_diag."""

    def __init__(self, loc=None, scale_diag=None, name='MultivariateNormalDiag'):
        self.loc, self.scale_diag, self.name = _t(loc), _t(scale_diag), name

    @property
    def event_shape(self):
        return torch.broadcast_shapes(self.loc.shape, self.scale_diag.shape)[-1:]

    @property
    def batch_shape(self):
        return torch.broadcast_shapes(self.loc.shape, self.scale_diag.shape)[:-1]

    def log_prob(self, x):
        sc = _on(self.scale_diag, x.device)
        z = (x - _on(self.loc, x.device)) / sc
        return (-0.5 * z * z - 0.5 * LN2PI - torch.log(sc.abs())).sum(-1)

    log_prob_flat = log_prob

    def score_flat(self, x):
        sc = _on(self.scale_diag, x.device)
        return -(x - _on(self.loc, x.device)) / (sc * sc)

    def sample(self, sample_shape=(), generator=None):
        shape = tuple(sample_shape) + tuple(torch.broadcast_shapes(self.loc.shape, self.scale_diag.shape))
        eps = torch.randn(shape, generator=generator).to(self.loc.device)
        return self.loc + self.scale_diag * eps


class WishartTriL(Distribution):
    """tfd.WishartTriL(df, scale_tril, input_output_cholesky=True): a Wishart(df, V = S S^T) matrix handled through
    its Cholesky factor L (exploratory_experiments/multivariate-gaussian-test.ipynb cell 15).  `log_prob(L)` is the
    Wishart log density of L L^T -- TFP adds no Jacobian term for L -> L L^T -- and `sample` returns Cholesky
    factors.  Only the Cholesky form is implemented."""

    def __init__(self, df, scale_tril, input_output_cholesky=False, name='WishartTriL'):
        if not input_output_cholesky:
            raise NotImplementedError('WishartTriL is implemented for input_output_cholesky=True only')
        self.df, self.name = float(df), name
        self.scale_tril = torch.tril(_t(scale_tril).to(torch.float32))
        self.d = int(self.scale_tril.shape[-1])
        if self.scale_tril.dim() != 2 or self.df <= self.d - 1:
            raise ValueError('WishartTriL needs one [D, D] scale_tril and df > D - 1')
        s64 = self.scale_tril.double()
        self._vinv = torch.cholesky_inverse(s64).float()                      # (S S^T)^-1
        d, df = self.d, self.df
        self._lognorm = float(0.5 * df * d * math.log(2.0) + df * torch.log(torch.diagonal(s64)).sum()
                              + 0.25 * d * (d - 1) * math.log(math.pi)
                              + sum(math.lgamma(0.5 * (df - m)) for m in range(d)))

    @property
    def event_shape(self):
        return torch.Size((self.d, self.d))

    def isotropic_scale(self):
        """s if scale_tril = s I, else None."""
        s = float(self.scale_tril[0, 0])
        return s if s > 0 and bool((self.scale_tril == s * torch.eye(self.d)).all()) else None

    def log_prob(self, x):
        return self.log_prob_flat(x.reshape(x.shape[:-2] + (self.d * self.d,)))

    def log_prob_flat(self, x):
        d = self.d
        L = torch.tril(x.reshape(x.shape[:-1] + (d, d)))
        vinv = _on(self._vinv, x.device).to(x.dtype)
        tr = ((vinv @ L) * L).sum((-2, -1))
        half_log_det = torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1)
        return (self.df - d - 1.0) * half_log_det - 0.5 * tr - self._lognorm

    def score_flat(self, x):
        d = self.d
        L = torch.tril(x.reshape(x.shape[:-1] + (d, d)))
        vinv = _on(self._vinv, x.device).to(x.dtype)
        g = -(vinv @ L) + torch.diag_embed((self.df - d - 1.0) / torch.diagonal(L, dim1=-2, dim2=-1))
        return torch.tril(g).reshape(x.shape)

    def sample(self, sample_shape=(), generator=None):
        """Bartlett: L = S A, A lower triangular with A_ii^2 ~ chi^2(df - i) and standard normals below."""
        shape = tuple(sample_shape)
        d = self.d
        a = torch.tril(torch.randn(shape + (d, d), generator=generator), -1)
        conc = torch.tensor([0.5 * (self.df - i) for i in range(d)]).expand(shape + (d,))
        chi = torch.sqrt(2.0 * torch._standard_gamma(conc))
        return self.scale_tril @ (a + torch.diag_embed(chi))

    def default_bijector(self):
        return tfb.FillScaleTriL(self.d)


class MixtureSameFamily(Distribution):
    def __init__(self, mixture_distribution, components_distribution, name='MixtureSameFamily'):
        self.mixture_distribution = mixture_distribution
        self.components_distribution = components_distribution
        self.name = name

    def sample(self, sample_shape=(), generator=None):
        """Ancestral sampling for concrete parameters: a component index per draw (weights normalised
        by their sum, as Categorical(probs=...) does), then that component's draw.  The component
        axis is the last batch axis of the component distribution."""
        mix = self.mixture_distribution
        probs = mix.probs if mix.probs is not None else torch.softmax(mix.logits, -1)
        probs = probs / probs.sum(-1, keepdim=True)
        comp = self.components_distribution.sample(sample_shape, generator=generator)   # [..., K(, D)]
        k = probs.shape[-1]
        event_dims = comp.dim() - len(tuple(sample_shape)) - probs.dim()
        lead = comp.shape[:comp.dim() - event_dims - 1]
        flat_p = probs.expand(tuple(lead) + (k,)).reshape(-1, k)
        idx = torch.multinomial(flat_p.cpu(), 1, generator=generator).reshape(tuple(lead)).to(comp.device)
        idx = idx.reshape(tuple(lead) + (1,) * (event_dims + 1)).expand(tuple(lead) + (1,) + tuple(comp.shape[comp.dim() - event_dims:]))
        return comp.gather(len(lead), idx).squeeze(len(lead))


class Sample(Distribution):
    """tfd.Sample(distribution, sample_shape): iid product (bayes_vi/model/model.py:17-21)."""

    def __init__(self, distribution, sample_shape=(), name='Sample'):
        self.distribution, self.sample_shape, self.name = distribution, sample_shape, name

    def sample(self, sample_shape=(), generator=None):
        n = self.sample_shape if isinstance(self.sample_shape, (tuple, list)) else (int(self.sample_shape),)
        draws = self.distribution.sample(tuple(n) + tuple(sample_shape), generator=generator)
        # iid axes go after the batch axes of the draw: [n..., sample..., batch..., event...] -> [sample..., batch..., n..., event...]
        return draws
