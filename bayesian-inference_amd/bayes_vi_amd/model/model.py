"""`Model`: priors + likelihood + constraining bijectors, and its joint log density.

Host-side mirror of bayes_vi/model/model.py.  Same constructor and attribute
names (model.py:85-169); the joint `log_prob` that the reference delegates to
`tfd.JointDistributionNamedAutoBatched` (model.py:171-203) is assembled here as
sum of log-priors (torch, O(#parameters)) plus the likelihood summed over every
datapoint, which is recognised symbolically (`bayes_vi_amd.symbolic`) and
routed to the fused HIP kernels (`bayes_vi_amd.fused_target`).
"""
from __future__ import annotations

import collections
import inspect
import math

import torch

from .. import bijectors as tfb
from .. import distributions as tfd
from .. import symbolic as sym
from ..fused_target import (FusedLogLikelihood, GaussianMixture1D, GaussianMixtureFullCov, LinearRegression,
                            MultivariateNormalTriL, NormalIID, Posterior, ShardedLogLikelihood)
from ..utils import to_ordered_dict


def _prod(shape):
    return int(math.prod(shape)) if len(shape) else 1


class _MappedLikelihood(ShardedLogLikelihood):
    """Adapts a fused family's parameter layout to the model's flat layout:
    kernel_theta[:, j] = model_theta[:, index[j]] (or the constant 0 where index[j] < 0)."""

    def __init__(self, inner, index, n_model_params):
        self.inner = inner
        self.n_params = n_model_params
        self.device = inner.device
        self.group, self.sharded = inner.group, inner.sharded
        idx = torch.as_tensor(index, dtype=torch.long)
        self.identity = bool((idx == torch.arange(n_model_params)).all()) if len(idx) == n_model_params else False
        self.valid = (idx >= 0).to(self.device)
        self.gather = idx.clamp_min(0).to(self.device)
        self.vidx = torch.nonzero(idx >= 0).reshape(-1).to(self.device)
        self.target_cols = idx[idx >= 0].to(self.device)

    @property
    def logp_ref(self):
        """The level the wrapped kernels subtract from their log-likelihoods (see FusedLogLikelihood)."""
        return self.inner.logp_ref

    @logp_ref.setter
    def logp_ref(self, value):
        self.inner.logp_ref = value

    def logp_grad(self, theta, want_grad=True, relative=False):
        if self.identity:
            return self.inner.logp_grad(theta, want_grad, relative)
        tk = theta.index_select(1, self.gather) * self.valid.to(theta.dtype)
        lp, gk = self.inner.logp_grad(tk.contiguous(), want_grad, relative)
        if not want_grad:
            return lp, None
        g = torch.zeros_like(theta)
        g.index_add_(1, self.target_cols, gk.index_select(1, self.vidx))
        return lp, g


def sample(f, num_samples=1):
    """Maps the data distribution onto the product distribution over `num_samples` iid draws
    (the `sample` decorator of bayes_vi/model/model.py:17-21), keeping the argument names."""
    def wrapped(*args, **kwargs):
        return tfd.Sample(f(*args, **kwargs), sample_shape=num_samples, name='Sample_Likelihood')
    wrapped.__signature__ = inspect.signature(f)
    wrapped.__name__ = getattr(f, '__name__', 'likelihood')
    return wrapped


class JointDistribution:
    """The joint of priors and likelihood for a fixed feature batch: what
    `Model.get_joint_distribution` returns (model.py:236-251).  `log_prob(**params, y=targets)`
    is the closure body of mcmc.py:57-59; `posterior(targets)` exposes the fused object
    (value + gradient, constrained and unconstrained) the samplers use directly."""

    def __init__(self, model, num_samples=None, features=None, group=None, precision=None):
        self.model, self.num_samples, self.features = model, num_samples, features
        self.group, self.precision = group, precision
        self._cache = None

    def posterior(self, targets) -> Posterior:
        if self._cache is None or self._cache[0] is not targets:
            lik = self.model._build_likelihood(self.features, targets, self.group, self.precision)
            post = Posterior(lik, [(p, _prod(s)) for p, s in zip(self.model.priors.values(),
                                                                   self.model.param_event_shape)],
                             self.model.constraining_bijectors)
            self._cache = (targets, post)
        return self._cache[1]

    def log_prob(self, *args, **kwargs):
        if args:
            kwargs = dict(args[0], **kwargs) if isinstance(args[0], dict) else kwargs
        y = kwargs.pop('y')
        parts = [kwargs[k] for k in self.model.param_names]
        theta = self.model.merge_state_parts(parts)
        return self.posterior(y).log_prob(theta)


class Model:
    """A probabilistic model in the Bayesian sense (see bayes_vi/model/model.py:24-83 for
    the attribute list this class reproduces)."""

    def __init__(self, priors, likelihood, constraining_bijectors=None):
        self.likelihood = likelihood
        self.is_generative_model = 'features' not in inspect.signature(likelihood).parameters.keys()

        self.param_names = list(priors.keys())
        self.priors = collections.OrderedDict(priors)
        self.dtypes = collections.OrderedDict([(k, v.dtype) for k, v in priors.items()])
        self.constraining_bijectors = constraining_bijectors
        if not self.constraining_bijectors:
            self.constraining_bijectors = [p.default_bijector() for p in self.priors.values()]
        self.joint_constraining_bijector = self.constraining_bijectors   # applied part-wise

        self.param_event_shape = [p.param_event_shape() for p in self.priors.values()]
        for b, shape in zip(self.constraining_bijectors, self.param_event_shape):
            if hasattr(b, 'bind'):
                b.bind(shape)
        block_sizes = [_prod(s) for s in self.param_event_shape]
        unconstrained_block_sizes = [b.inverse_event_size(n) for b, n in zip(self.constraining_bijectors, block_sizes)]
        self.unconstrained_param_event_shape = [
            torch.Size(s) if u == n else torch.Size((u,))
            for s, n, u in zip(self.param_event_shape, block_sizes, unconstrained_block_sizes)]
        self.flat_param_event_shape = [torch.Size((n,)) for n in block_sizes]
        self.flat_unconstrained_param_event_shape = [torch.Size((n,)) for n in unconstrained_block_sizes]
        self.block_sizes, self.unconstrained_block_sizes = block_sizes, unconstrained_block_sizes

        self.blockwise_constraining_bijector = tfb.Blockwise(self.constraining_bijectors, unconstrained_block_sizes)
        self.flat_param_event_ndims = sum(block_sizes)
        self.flat_unconstrained_param_event_ndims = sum(unconstrained_block_sizes)

    # ---- flat <-> parts (split_flat_param_bijector / reshape_flat_param_bijector, model.py:123-153) ----
    def split_state(self, flat, unconstrained=False):
        """flat[..., P] -> list of parts with their event shapes (leading dims kept)."""
        sizes = self.unconstrained_block_sizes if unconstrained else self.block_sizes
        shapes = self.unconstrained_param_event_shape if unconstrained else self.param_event_shape
        parts = torch.split(flat, sizes, dim=-1)
        return [p.reshape(p.shape[:-1] + tuple(s)) for p, s in zip(parts, shapes)]

    def merge_state_parts(self, parts, unconstrained=False):
        """list of parts [..., *event] -> flat[..., P] (param_names order, row-major per part)."""
        shapes = self.unconstrained_param_event_shape if unconstrained else self.param_event_shape
        flat = []
        for p, s in zip(parts, shapes):
            p = torch.as_tensor(p, dtype=torch.float32)
            lead = p.shape[:p.dim() - len(s)]
            flat.append(p.reshape(tuple(lead) + (_prod(s),)))
        return torch.cat(flat, -1)

    # ---- likelihood recognition ----
    def _offsets(self):
        off, o = {}, 0
        for name, n in zip(self.param_names, self.block_sizes):
            off[name] = o
            o += n
        return off

    def describe_likelihood(self, features=None):
        """Symbolic recognition only (no data touched, no GPU): returns a dict with
        'family' in {'normal_iid', 'linreg', 'gmm1d', 'gmm_fullcov', 'mvn_tril'}, 'index' (kernel parameter j <- model flat
        index, -1 = constant 0) and, for 'linreg', the design-column recipe:
        'columns' = [[(feature_key, coef), ...] per slope parameter] and 'offset' =
        [(feature_key | None, coef), ...] terms to subtract from y."""
        syms = {name: sym.ParamSym(name, n) for name, n in zip(self.param_names, self.block_sizes)}
        kwargs = dict(syms)
        if not self.is_generative_model:
            if isinstance(features, dict):
                kwargs['features'] = sym.FeatureSet(list(features.keys()), True)
            else:
                f = torch.as_tensor(features)
                kwargs['features'] = sym.FeatureSet(list(range(1 if f.dim() == 1 else f.shape[1])), False)
        dist = self.likelihood(**kwargs)
        if isinstance(dist, tfd.Sample):
            dist = dist.distribution
        off = self._offsets()

        def pidx(ref):
            name, j = ref
            return off[name] + (0 if j is None else j)

        if isinstance(dist, tfd.Normal):
            scale = dist.scale
            if not sym.is_plain_param(scale, 1):
                raise sym.UnsupportedLikelihood('Normal likelihood: `scale` must be one scalar parameter')
            if sym.is_plain_param(dist.loc, 1) and self.is_generative_model:
                return dict(family='normal_iid', index=[off[dist.loc.name], off[scale.name]])
            if self.is_generative_model:
                raise sym.UnsupportedLikelihood('a generative Normal likelihood must have loc = one scalar parameter')
            loc = sym.Affine.lift(dist.loc)
            intercept, order, columns, offset = None, [], {}, []
            for p, f, c in loc.terms:
                if p is None:
                    offset.append((f, c))
                elif f is None:
                    if c != 1.0 or intercept is not None:
                        raise sym.UnsupportedLikelihood('at most one intercept parameter with coefficient 1 is fused')
                    intercept = p
                else:
                    if p not in columns:
                        columns[p] = []
                        order.append(p)
                    columns[p].append((f, c))
            if not order:
                raise sym.UnsupportedLikelihood('regression likelihood uses no feature')
            index = [pidx(intercept) if intercept is not None else -1] + [pidx(p) for p in order] + [off[scale.name]]
            return dict(family='linreg', index=index, columns=[columns[p] for p in order], offset=offset)
        if isinstance(dist, tfd.MixtureSameFamily):
            mix, comp = dist.mixture_distribution, dist.components_distribution
            if not isinstance(mix, tfd.Categorical) or mix.probs is None or not sym.is_plain_param(mix.probs):
                raise sym.UnsupportedLikelihood('mixture weights must be Categorical(probs=<one parameter>)')
            k = mix.probs.size
            if isinstance(comp, tfd.Normal) and sym.is_plain_param(comp.loc, k) and sym.is_plain_param(comp.scale, k):
                index = ([off[mix.probs.name] + j for j in range(k)] + [off[comp.loc.name] + j for j in range(k)]
                         + [off[comp.scale.name] + j for j in range(k)])
                return dict(family='gmm1d', k=k, index=index)
            if isinstance(comp, tfd.MultivariateNormalTriL) and sym.is_plain_param(comp.loc) \
                    and sym.is_plain_param(comp.scale_tril) and comp.loc.size % k == 0:
                d = comp.loc.size // k
                if comp.scale_tril.size != k * d * d:
                    raise sym.UnsupportedLikelihood('scale_tril must have shape [K, D, D] with D = %d' % d)
                index = ([off[mix.probs.name] + j for j in range(k)]
                         + [off[comp.loc.name] + j for j in range(k * d)]
                         + [off[comp.scale_tril.name] + j for j in range(k * d * d)])
                return dict(family='gmm_fullcov', k=k, d=d, index=index)
            raise sym.UnsupportedLikelihood('only mixtures of Normal(loc[K], scale[K]) or of '
                                            'MultivariateNormalTriL(loc[K, D], scale_tril[K, D, D]) are fused')
        if isinstance(dist, tfd.MultivariateNormalTriL):
            if not sym.is_plain_param(dist.loc):
                raise sym.UnsupportedLikelihood('MultivariateNormalTriL likelihood: `loc` must be one vector parameter')
            d = dist.loc.size
            mat, mode = dist.scale_tril, 'scale_tril'
            if isinstance(mat, sym.CholeskySolveIdentity):
                mat, mode = mat.param, 'cholesky_solve_identity'
            if not sym.is_plain_param(mat, d * d) or not self.is_generative_model:
                raise sym.UnsupportedLikelihood('MultivariateNormalTriL likelihood: scale_tril must be one [D, D] '
                                                'parameter or linalg.cholesky_solve(<[D, D] parameter>, linalg.eye(D))')
            index = [off[dist.loc.name] + j for j in range(d)] + [off[mat.name] + j for j in range(d * d)]
            return dict(family='mvn_tril', d=d, parameterisation=mode, index=index)
        raise sym.UnsupportedLikelihood('likelihood %s is not one of the fused families '
                                        '(Normal, Normal linear regression, MixtureSameFamily of Normal or of '
                                        'MultivariateNormalTriL, MultivariateNormalTriL)'
                                        % type(dist).__name__)

    def design_matrix(self, desc, features, targets):
        """Materialises (X[N, D], y_adjusted[N]) of a 'linreg' description."""
        if isinstance(features, dict):
            col = lambda k: torch.as_tensor(features[k], dtype=torch.float32).reshape(-1)
        else:
            fmat = torch.as_tensor(features, dtype=torch.float32)
            fmat = fmat[:, None] if fmat.dim() == 1 else fmat
            col = lambda k: fmat[:, k]
        y = torch.as_tensor(targets, dtype=torch.float32).reshape(-1).clone()
        for f, c in desc['offset']:
            y = y - (c if f is None else c * col(f).to(y.device))
        X = torch.stack([sum(c * col(f) for f, c in terms) for terms in desc['columns']], -1)
        return X, y

    @staticmethod
    def _to_device(t):
        if t.is_cuda:
            return t
        if not torch.cuda.is_available():
            from .._native import BviError
            raise BviError('no GPU visible: the fused log-likelihood kernels run on MI355X only')
        return t.cuda()

    def _build_likelihood(self, features, targets, group=None, precision=None) -> FusedLogLikelihood:
        from .. import _native
        desc = self.describe_likelihood(features)
        P = self.flat_param_event_ndims
        y = torch.as_tensor(targets, dtype=torch.float32)
        if desc['family'] == 'normal_iid':
            inner = NormalIID(self._to_device(y), group=group)
        elif desc['family'] == 'gmm1d':
            inner = GaussianMixture1D(self._to_device(y), desc['k'], group=group)
        elif desc['family'] == 'gmm_fullcov':
            inner = GaussianMixtureFullCov(self._to_device(y.reshape(-1, desc['d'])), desc['k'], group=group)
        elif desc['family'] == 'mvn_tril':
            mode = _native.MVN_SCALE_TRIL if desc['parameterisation'] == 'scale_tril' else _native.MVN_CHOL_PRECISION_SOLVE
            inner = MultivariateNormalTriL(self._to_device(y.reshape(-1, desc['d'])), mode, group=group)
        else:
            X, y_adj = self.design_matrix(desc, features, y)
            inner = LinearRegression(self._to_device(X), self._to_device(y_adj), group=group,
                                     precision=_native.PREC_AUTO if precision is None else precision)
        return _MappedLikelihood(inner, desc['index'], P)

    # ---- joint distributions (model.py:171-277) ----
    def get_joint_distribution(self, num_samples=None, features=None, group=None, precision=None):
        """The joint of the model conditioned on `features` (regression) or over `num_samples`
        iid observations (generative): model.py:236-251."""
        return JointDistribution(self, num_samples=num_samples, features=features, group=group, precision=precision)

    def get_param_distributions(self, joint_param_distribution=None, param_samples=None):
        """Marginal parameter distributions from samples, as empirical sample sets (model.py:205-234)."""
        if isinstance(param_samples, (list, collections.OrderedDict)):
            if isinstance(param_samples, list):
                param_samples = to_ordered_dict(self.param_names, param_samples)
            return collections.OrderedDict(
                (name, Empirical(part.reshape((-1,) + tuple(shape))))
                for (name, part), shape in zip(param_samples.items(), self.param_event_shape))
        raise ValueError('You have to provide either a joint distribution or param samples.')

    def get_posterior_predictive_distribution(self, posterior_distribution=None, posterior_samples=None,
                                              num_samples=None, features=None):
        """The joint of (empirical) posterior marginals and the likelihood, for drawing replicated data
        (model.py:253-277): `.sample(sample_shape)` returns an OrderedDict of the parameter draws and 'y'."""
        posteriors = posterior_distribution if isinstance(posterior_distribution, collections.OrderedDict) \
            else self.get_param_distributions(param_samples=posterior_samples)
        return PredictiveDistribution(self, posteriors, num_samples=num_samples, features=features)

    def get_prior_predictive_distribution(self, num_samples=None, features=None):
        """The generative joint itself: priors and likelihood (what `get_joint_distribution(...).sample` gives
        in the reference, model.py:236-251)."""
        return PredictiveDistribution(self, self.priors, num_samples=num_samples, features=features)

    def sample_prior(self, sample_shape=(), generator=None):
        return [p.sample(sample_shape, generator=generator) for p in self.priors.values()]


class Empirical:
    """Sample-based marginal (stands in for tfd.Empirical, model.py:228)."""

    def __init__(self, samples):
        self.samples = samples

    def mean(self):
        return self.samples.mean(0)

    def stddev(self):
        return self.samples.std(0, unbiased=False)

    def sample(self, sample_shape=(), generator=None):
        n = int(math.prod(sample_shape)) if len(sample_shape) else 1
        idx = torch.randint(0, self.samples.shape[0], (n,), generator=generator)
        out = self.samples[idx.to(self.samples.device)]
        return out.reshape(tuple(sample_shape) + tuple(self.samples.shape[1:]))


class PredictiveDistribution:
    """Ancestral sampler of a joint `parameters -> y`: parameter marginals (priors or empirical
    posteriors) and the model's likelihood evaluated on concrete draws, conditioned on `features` for
    regression models or over `num_samples` iid observations for generative ones (model.py:171-203)."""

    def __init__(self, model, param_distributions, num_samples=None, features=None):
        self.model, self.param_distributions = model, param_distributions
        self.num_samples, self.features = num_samples, features

    def sample(self, sample_shape=(), generator=None):
        sample_shape = (sample_shape,) if isinstance(sample_shape, int) else tuple(sample_shape)
        draws = collections.OrderedDict(
            (name, dist.sample(sample_shape, generator=generator)) for name, dist in self.param_distributions.items())
        m = self.model
        if m.is_generative_model:
            lik = m.likelihood(**draws)
            n = int(self.num_samples or 1)
            y = lik.sample((n,), generator=generator)                    # [n, *sample_shape, ...]
            y = y.movedim(0, len(sample_shape))                          # iid axis after the draw axes
        else:
            feats = self.features
            # per-row location: parameters gain a trailing row axis where the likelihood broadcasts over rows
            expanded = collections.OrderedDict(
                (k, v[..., None] if v.dim() == len(sample_shape) else v) for k, v in draws.items())
            lik = m.likelihood(**expanded, features=feats)
            y = lik.sample((), generator=generator)
        out = collections.OrderedDict(draws)
        out['y'] = y
        return out
