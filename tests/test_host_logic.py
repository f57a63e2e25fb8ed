"""CPU tests of the host side: likelihood recognition, bijector plumbing, samplers
(driven by the fp64 oracle as the target), step-size adaptation, leapfrog integrator."""
import collections
import math

import numpy as np
import pytest
import torch

from bayes_vi_amd import bijectors as tfb
from bayes_vi_amd import distributions as tfd
from bayes_vi_amd import symbolic as sym
from bayes_vi_amd.inference.mcmc import samplers
from bayes_vi_amd.inference.mcmc.stepsize_adaptation_kernels import DualAveragingStepSizeAdaptation
from bayes_vi_amd.inference.mcmc.transition_kernels import (HamiltonianMonteCarlo, NoUTurnSampler,
                                                             RandomWalkMetropolis)
from bayes_vi_amd.model import Model
from bayes_vi_amd.utils import compose, to_ordered_dict
from bayes_vi_amd.utils.symplectic_integrators import LeapfrogIntegrator
from oracle import closed_form as cf
from oracle import leapfrog as lf


def oracle_target(spec):
    def target(z, want_grad):
        lp, g = spec.log_post_unconstrained(z.detach().cpu().numpy().astype(np.float64))
        return torch.tensor(lp, dtype=z.dtype), (torch.tensor(g, dtype=z.dtype) if want_grad else None)
    return target


# ---------------------------------------------------------------- model / recognition
def normal_model():
    priors = collections.OrderedDict(loc=tfd.Normal(loc=0., scale=10.), scale=tfd.HalfNormal(scale=10.))
    return Model(priors=priors, likelihood=lambda loc, scale: tfd.Normal(loc=loc, scale=scale))


def test_model_attributes_normal():
    m = normal_model()
    assert m.is_generative_model and m.param_names == ['loc', 'scale']
    assert m.flat_param_event_ndims == 2 and m.flat_unconstrained_param_event_ndims == 2
    assert isinstance(m.constraining_bijectors[0], tfb.Identity) and isinstance(m.constraining_bijectors[1], tfb.Softplus)
    d = m.describe_likelihood()
    assert d['family'] == 'normal_iid' and d['index'] == [0, 1]
    # order of the priors decides the flat layout, the kernel layout is fixed
    m2 = Model(collections.OrderedDict(scale=tfd.HalfNormal(10.), loc=tfd.Normal(0., 10.)),
               lambda loc, scale: tfd.Normal(loc=loc, scale=scale))
    assert m2.describe_likelihood()['index'] == [1, 0]


def test_recognise_readme_regression():
    priors = collections.OrderedDict(beta_0=tfd.Normal(0., 10.), beta_1=tfd.Normal(0., 10.), scale=tfd.HalfNormal(10.))

    def likelihood(beta_0, beta_1, scale, features):
        linear_response = beta_0 + beta_1 * features['x']
        return tfd.Normal(loc=linear_response, scale=scale)
    m = Model(priors, likelihood)
    assert not m.is_generative_model
    feats = {'x': torch.arange(5.)}
    d = m.describe_likelihood(feats)
    assert d['family'] == 'linreg' and d['index'] == [0, 1, 2] and d['columns'] == [[('x', 1.0)]]
    X, y = m.design_matrix(d, feats, torch.ones(5))
    assert X.shape == (5, 1) and torch.equal(X[:, 0], feats['x'])


def test_recognise_vector_regression_and_offsets():
    dfeat = 4
    priors = collections.OrderedDict(w=tfd.Normal(torch.zeros(dfeat), 10.), b=tfd.Normal(0., 10.), s=tfd.HalfNormal(10.))

    def likelihood(w, b, s, features):
        return tfd.Normal(loc=sym.matvec(features, w) + b + 2.0, scale=s)
    m = Model(priors, likelihood)
    assert m.flat_param_event_ndims == 6
    F = torch.randn(7, dfeat)
    d = m.describe_likelihood(F)
    assert d['index'] == [4, 0, 1, 2, 3, 5]
    X, y = m.design_matrix(d, F, torch.zeros(7))
    assert torch.allclose(X, F) and torch.allclose(y, torch.full((7,), -2.0))

    def bad(w, b, s, features):
        return tfd.Normal(loc=b * b, scale=s)
    with pytest.raises(sym.UnsupportedLikelihood):
        Model(priors, bad).describe_likelihood(F)


def test_recognise_mixture():
    k = 3
    priors = collections.OrderedDict(weights=tfd.Dirichlet(torch.ones(k)), loc=tfd.Normal(torch.zeros(k), 10.),
                                     scale=tfd.HalfNormal(torch.ones(k) * 10.))

    def likelihood(weights, loc, scale):
        return tfd.MixtureSameFamily(mixture_distribution=tfd.Categorical(probs=weights),
                                     components_distribution=tfd.Normal(loc=loc, scale=scale))
    m = Model(priors, likelihood)
    assert m.flat_param_event_ndims == 9 and m.flat_unconstrained_param_event_ndims == 8
    assert [tuple(s) for s in m.unconstrained_param_event_shape] == [(2,), (3,), (3,)]
    d = m.describe_likelihood()
    assert d['family'] == 'gmm1d' and d['k'] == 3 and d['index'] == list(range(9))
    z = torch.randn(5, 8)
    th = m.blockwise_constraining_bijector.forward(z)
    assert th.shape == (5, 9) and torch.allclose(th[:, :3].sum(-1), torch.ones(5), atol=1e-6)
    parts = m.split_state(th)
    assert [tuple(p.shape) for p in parts] == [(5, 3), (5, 3), (5, 3)]
    assert torch.equal(m.merge_state_parts(parts), th)


@pytest.mark.parametrize('name', ['Identity', 'Softplus', 'Exp', 'SoftmaxCentered', 'FillScaleTriL'])
def test_torch_bijectors_match_oracle(name):
    rng = np.random.default_rng(0)
    if name == 'FillScaleTriL':
        b, o, n = tfb.FillScaleTriL(3, 1e-3), cf.FillScaleTriL(3, 1e-3), 6
    else:
        b, o, n = getattr(tfb, name)(), getattr(cf, name)(), 4
    x = rng.normal(size=(3, n))
    xt = torch.tensor(x)
    np.testing.assert_allclose(b.forward(xt).numpy(), o.forward(x), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(b.forward_log_det_jacobian(xt).numpy(), o.fldj(x), rtol=1e-12, atol=1e-12)
    g = rng.normal(size=o.forward(x).shape)
    np.testing.assert_allclose(b.vjp(xt, torch.tensor(g)).numpy(), o.vjp(x, g), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(b.grad_fldj(xt).numpy(), o.grad_fldj(x), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(b.inverse(b.forward(xt)).numpy(), x, rtol=1e-6, atol=1e-6)


def test_priors_match_oracle():
    x = torch.tensor([[0.3, -1.2, 2.0]], dtype=torch.float64)
    p = tfd.Normal(torch.zeros(3), 10.)
    lp, g = cf.normal_prior(x.numpy(), 0.0, 10.0)
    np.testing.assert_allclose(p.log_prob_flat(x).numpy(), lp.sum(-1), rtol=1e-6)
    np.testing.assert_allclose(p.score_flat(x).numpy(), g, rtol=1e-6)
    h = tfd.HalfNormal(10.)
    lp, g = cf.halfnormal_prior(np.array([[2.4]]), 10.0)
    np.testing.assert_allclose(h.log_prob_flat(torch.tensor([[2.4]])).numpy(), lp.sum(-1), rtol=1e-6)
    d = tfd.Dirichlet(torch.ones(3))
    assert float(d.log_prob_flat(torch.tensor([[0.2, 0.3, 0.5]]))) == pytest.approx(math.lgamma(3), rel=1e-6)


def test_utils():
    f = compose([lambda a, b: (a + b, a), lambda s, a: (s * 2, a)])
    assert f(1, 2) == (6, 1)
    od = to_ordered_dict(['a', 'b'], [1, 2])
    assert list(od.items()) == [('a', 1), ('b', 2)]


# ---------------------------------------------------------------- leapfrog integrator
def test_leapfrog_integrator_readme_orbit():
    T = lambda p: 0.5 * p ** 2
    U = lambda q: 0.5 * q ** 2
    integ = LeapfrogIntegrator()
    x = torch.tensor([1.0, 0.0], dtype=torch.float64)
    path = [x]
    for _ in range(20):
        path.append(integ.solve(U, T, path[-1], 0.5, 1))
    path = torch.stack(path).numpy()
    np.testing.assert_array_equal(path[1], [0.875, -0.46875])
    np.testing.assert_array_equal(path[2], [0.53125, -0.8203125])
    np.testing.assert_allclose(path[20], [-0.7760410416, 0.6106556172], atol=1e-9)
    np.testing.assert_allclose(integ.solve(U, T, torch.tensor(path[5]), -0.5, 1).numpy(), path[4], atol=1e-14)
    with pytest.raises(ValueError):
        integ.solve(U, T, torch.zeros(3), 0.1, 1)
    # batched + per-dimension, trainable step sizes keep gradients
    eps = torch.tensor([0.1, 0.2], dtype=torch.float64, requires_grad=True)
    out = integ.solve(U, T, torch.ones(4, 4, dtype=torch.float64), eps, 3)
    ref = lf.solve(lambda q: q, lambda p: p, np.ones((4, 4)), np.array([0.1, 0.2]), 3)
    np.testing.assert_allclose(out.detach().numpy(), ref, atol=1e-14)
    out.sum().backward()
    assert eps.grad is not None and torch.isfinite(eps.grad).all()


# ---------------------------------------------------------------- samplers
def test_hmc_matches_oracle_with_injected_randomness():
    rng = np.random.default_rng(1)
    spec = cf.normal_model(rng.normal(7, 3, 20))
    z = torch.tensor(rng.uniform(-1, 1, (4, 2)) + np.array([6.0, 1.0]))
    p = rng.normal(size=(4, 2)); lu = np.log(rng.uniform(size=4))
    k = samplers.HamiltonianMonteCarloKernel(oracle_target(spec), step_size=0.01, num_leapfrog_steps=5)
    st = k.bootstrap(z)
    z1, st1, tr = k.one_step(z, st, momenta=torch.tensor(p), log_uniform=torch.tensor(lu))
    q_ref, acc_ref, lr_ref, lp_ref = lf.hmc_step(spec.log_post_unconstrained, z.numpy(), p, lu, 0.01, 5)
    np.testing.assert_allclose(z1.numpy(), q_ref, atol=1e-12)
    np.testing.assert_array_equal(tr['is_accepted'].numpy(), acc_ref)
    np.testing.assert_allclose(tr['log_accept_ratio'].numpy(), lr_ref, atol=1e-10)
    np.testing.assert_allclose(st1['target_log_prob'].numpy(), lp_ref, atol=1e-10)


def test_nonfinite_proposals_are_rejected():
    def target(z, want_grad):
        lp = torch.where(z[:, 0] > 0.5, torch.full_like(z[:, 0], float('nan')), -0.5 * (z * z).sum(-1))
        return lp, (-z if want_grad else None)
    g = torch.Generator().manual_seed(0)
    k = samplers.RandomWalkMetropolisKernel(target)
    z = torch.zeros(64, 1, dtype=torch.float64)
    st = k.bootstrap(z)
    for _ in range(20):
        z, st, tr = k.one_step(z, st, generator=g)
        assert torch.isfinite(st['target_log_prob']).all() and (z <= 0.5).all()


def test_dual_averaging_recursion():
    class Dummy:
        trace_names = ('is_accepted',)
        step_size = 0.01
        def bootstrap(self, z): return {}
        def one_step(self, z, state, generator=None):
            return z, state, {'log_accept_ratio': torch.log(torch.tensor([0.5, 1.0])), 'is_accepted': None}
    d = Dummy()
    ada = samplers.DualAveragingStepSizeAdaptation(d, num_adaptation_steps=3)
    mu = math.log(10 * 0.01)
    err, xbar = 0.0, 0.0
    for t in range(1, 4):
        ada.one_step(None, {})
        err += 0.75 - 0.75
        x = mu - math.sqrt(t) / 0.05 * err / (t + 10)
        eta = t ** -0.75
        xbar = eta * x + (1 - eta) * xbar
        expect = math.exp(x) if t < 3 else math.exp(xbar)
        assert d.step_size == pytest.approx(expect, rel=1e-12)
    ada.one_step(None, {})
    assert d.step_size == pytest.approx(math.exp(xbar), rel=1e-12)     # frozen after adaptation


@pytest.mark.parametrize('kernel_name', ['nuts', 'hmc', 'rwm'])
def test_posterior_moments_match_reference_table(kernel_name):
    """univariate_gaussian_experiment dataset 1 (dataset_infos.txt:6, mcmc_summary_statistics.txt:6-7):
    N=100, sample loc 3.37, sample scale 1.65 -> NUTS posterior loc 3.37 +- 0.16, scale 1.68 +- 0.12.
    The posterior depends on the data only through those sufficient statistics."""
    rng = np.random.default_rng(3)
    y = rng.normal(size=100); y = (y - y.mean()) / y.std(ddof=1) * 1.65 + 3.37
    spec = cf.normal_model(y)
    target = oracle_target(spec)
    gen = torch.Generator().manual_seed(11)
    c = 4
    z0 = torch.tensor(spec.unconstrain(np.array([[3.0, 2.0]] * c)) + rng.uniform(-.3, .3, (c, 2)))
    if kernel_name == 'nuts':
        cfg = NoUTurnSampler(stepsize_adaptation_kernel=DualAveragingStepSizeAdaptation(num_adaptation_steps=400))
        burn, draws = 500, 1000
    elif kernel_name == 'hmc':
        cfg = HamiltonianMonteCarlo(step_size=0.05, num_leapfrog_steps=8,
                                    stepsize_adaptation_kernel=DualAveragingStepSizeAdaptation(num_adaptation_steps=400))
        burn, draws = 500, 2000
    else:
        cfg = RandomWalkMetropolis(new_state_fn=lambda z, g: z + 0.15 * torch.randn(z.shape, generator=g, dtype=z.dtype))
        burn, draws = 1000, 6000
    # the reference's contract (transition_kernels.py:24-38, consumed at mcmc.py:150, 172)
    kernel, trace_fns, names = cfg(target)
    assert names == cfg.trace_fn_metrics and all(callable(f) for f in trace_fns)
    assert len(trace_fns) == (2 if kernel_name != 'rwm' else 1)       # own trace_fn + one unwrap per wrapping kernel
    from bayes_vi_amd.utils import compose
    zs, trace = samplers.sample_chain(kernel, z0, draws, burn, generator=gen,
                                      trace_fn=compose(list(reversed(trace_fns))))
    assert isinstance(trace, tuple) and len(trace) == len(names)
    tr = dict(zip(names, trace))
    assert all(v.shape[0] == draws for v in trace)
    th = spec.constrain(zs.reshape(-1, 2).numpy())
    tol = 0.03 if kernel_name != 'rwm' else 0.05
    assert th[:, 0].mean() == pytest.approx(3.37, abs=tol)
    assert th[:, 0].std() == pytest.approx(0.16, abs=0.02)
    assert th[:, 1].mean() == pytest.approx(1.68, abs=tol)
    assert th[:, 1].std() == pytest.approx(0.12, abs=0.02)
    acc = tr['is_accepted'].to(torch.float32).mean()
    assert 0.15 < float(acc) <= 1.0
    if kernel_name == 'nuts':
        assert int(tr['leapfrogs_taken'].max()) <= 31 and not bool(tr['has_divergence'].any())
        assert bool((tr['energy'] <= tr['target_log_prob'] + 1e-4).all())    # log density minus kinetic energy


def test_linreg_nuts_matches_reference_table():
    """linear_regression_experiment dataset 7 (dataset_infos.txt:12, mcmc_summary_statistics.txt:24-26): the
    reference's NUTS + dual averaging summary beta0 -8.12 +- 0.87, beta1 3.61 +- 0.09, scale 8.66 +- 0.63,
    sampled here by this repository's NUTS on the oracle's posterior of data standardised to the printed sample
    statistics (x itself was not saved: beta0 and the spreads carry that looseness)."""
    from test_oracle_pins import LINREG_TABLE, linreg_table_dataset
    (b0, b1, s), (m0, s0), (m1, s1), (ms, ss) = LINREG_TABLE[7]
    x, y = linreg_table_dataset(b0, b1, s)
    spec = cf.linreg_model(x[:, None], y)
    cfg = NoUTurnSampler(stepsize_adaptation_kernel=DualAveragingStepSizeAdaptation(num_adaptation_steps=400))
    kernel, trace_fns, names = cfg(oracle_target(spec))
    rng = np.random.default_rng(4)
    c = 4
    z0 = torch.tensor(spec.unconstrain(np.array([[b0, b1, s]] * c)) + rng.uniform(-.2, .2, (c, 3)))
    zs, tr = samplers.sample_chain(kernel, z0, 1500, 500, generator=torch.Generator().manual_seed(3))
    th = spec.constrain(zs.reshape(-1, 3).numpy())
    assert th[:, 1].mean() == pytest.approx(m1, abs=0.015)
    assert th[:, 2].mean() == pytest.approx(ms, abs=0.08)
    assert th[:, 2].std() == pytest.approx(ss, abs=0.06)
    assert th[:, 0].mean() == pytest.approx(m0, abs=0.15)
    assert th[:, 0].std() == pytest.approx(s0, rel=0.12)
    assert th[:, 1].std() == pytest.approx(s1, rel=0.3)
    assert not bool(tr['has_divergence'].any())


def test_recognise_fullcov_mixture_and_batched_tril():
    k, d = 3, 2
    m = d * (d + 1) // 2
    priors = collections.OrderedDict(weights=tfd.Dirichlet(torch.ones(k)),
                                     locs=tfd.Normal(torch.zeros(k, d), 10.),
                                     scale_tril=tfd.Normal(torch.zeros(k, d, d), 5.))

    def likelihood(weights, locs, scale_tril):
        return tfd.MixtureSameFamily(mixture_distribution=tfd.Categorical(probs=weights),
                                     components_distribution=tfd.MultivariateNormalTriL(loc=locs, scale_tril=scale_tril))
    bij = [tfb.SoftmaxCentered(), tfb.Identity(), tfb.FillScaleTriL(d, diag_shift=1e-3, batch=k)]
    model = Model(priors, likelihood, constraining_bijectors=bij)
    assert model.flat_param_event_ndims == k + k * d + k * d * d
    assert model.flat_unconstrained_param_event_ndims == (k - 1) + k * d + k * m
    desc = model.describe_likelihood()
    assert desc['family'] == 'gmm_fullcov' and desc['k'] == k and desc['d'] == d
    assert desc['index'] == list(range(k + k * d + k * d * d))
    # the batched bijector equals the oracle's single-block bijector applied per component
    rng = np.random.default_rng(0)
    x = rng.normal(size=(4, k * m))
    o = cf.FillScaleTriL(d, 1e-3)
    ref = np.concatenate([o.forward(x[:, i * m:(i + 1) * m]) for i in range(k)], -1)
    b = bij[2]
    np.testing.assert_allclose(b.forward(torch.tensor(x)).numpy(), ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(b.forward_log_det_jacobian(torch.tensor(x)).numpy(),
                               sum(o.fldj(x[:, i * m:(i + 1) * m]) for i in range(k)), rtol=1e-12)
    g = rng.normal(size=ref.shape)
    refv = np.concatenate([o.vjp(x[:, i * m:(i + 1) * m], g[:, i * d * d:(i + 1) * d * d]) for i in range(k)], -1)
    np.testing.assert_allclose(b.vjp(torch.tensor(x), torch.tensor(g)).numpy(), refv, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(b.inverse(b.forward(torch.tensor(x))).numpy(), x, rtol=1e-6, atol=1e-6)


def test_posterior_predictive_sampling():
    """model.py:253-277: replicated data from (empirical) posterior marginals and the likelihood, for a
    generative model, a dict-feature regression, a matvec regression and a mixture."""
    import bayes_vi_amd.distributions as tfd
    from bayes_vi_amd import symbolic
    from bayes_vi_amd.model import Model
    g = torch.Generator().manual_seed(0)
    priors = collections.OrderedDict(loc=tfd.Normal(0., 10.), scale=tfd.HalfNormal(10.))
    model = Model(priors, lambda loc, scale: tfd.Normal(loc=loc, scale=scale))
    post = [7 + 0.1 * torch.randn(500, 2, generator=g), 3 + 0.05 * torch.randn(500, 2, generator=g).abs()]
    pp = model.get_posterior_predictive_distribution(posterior_samples=post, num_samples=20)
    s = pp.sample((4000,), generator=g)
    assert list(s.keys()) == ['loc', 'scale', 'y'] and s['y'].shape == (4000, 20)
    assert abs(float(s['y'].mean()) - 7) < 0.1 and abs(float(s['y'].std()) - 3.03) < 0.1
    # regression, dict features
    x = torch.linspace(-3, 3, 50)
    pr = collections.OrderedDict(beta_0=tfd.Normal(0., 10.), beta_1=tfd.Normal(0., 10.), scale=tfd.HalfNormal(10.))
    reg = Model(pr, lambda beta_0, beta_1, scale, features: tfd.Normal(loc=beta_0 + beta_1 * features['x'], scale=scale))
    post = [1 + 0.01 * torch.randn(300, generator=g), 2 + 0.01 * torch.randn(300, generator=g),
            0.5 + 0.01 * torch.rand(300, generator=g)]
    s = reg.get_posterior_predictive_distribution(posterior_samples=post, features={'x': x}).sample((2000,), generator=g)
    assert s['y'].shape == (2000, 50)
    assert torch.allclose(s['y'].mean(0), 1 + 2 * x, atol=0.08)
    # regression, feature matrix through matvec
    X = torch.randn(40, 3, generator=g); w = torch.tensor([1.0, -2.0, 0.5])
    pr = collections.OrderedDict(b=tfd.Normal(0., 10.), w=tfd.Normal(torch.zeros(3), 10.), scale=tfd.HalfNormal(10.))
    reg = Model(pr, lambda b, w, scale, features: tfd.Normal(loc=b + symbolic.matvec(features, w), scale=scale))
    post = [0.3 + 0.01 * torch.randn(300, generator=g), w + 0.01 * torch.randn(300, 3, generator=g),
            0.2 + 0.01 * torch.rand(300, generator=g)]
    s = reg.get_posterior_predictive_distribution(posterior_samples=post, features=X).sample((3000,), generator=g)
    assert s['y'].shape == (3000, 40) and torch.allclose(s['y'].mean(0), 0.3 + X @ w, atol=0.05)
    # mixture: prior predictive
    k = 3
    pm = collections.OrderedDict(weights=tfd.Dirichlet(torch.ones(k)), loc=tfd.Normal(torch.tensor([-10., 0., 10.]), 0.1),
                                 scale=tfd.HalfNormal(torch.ones(k) * 0.5))
    mix = Model(pm, lambda weights, loc, scale: tfd.MixtureSameFamily(
        mixture_distribution=tfd.Categorical(probs=weights), components_distribution=tfd.Normal(loc=loc, scale=scale)))
    s = mix.get_prior_predictive_distribution(num_samples=5).sample((3000,), generator=g)
    assert s['y'].shape == (3000, 5)
    near = (s['y'][..., None] - torch.tensor([-10., 0., 10.])).abs().min(-1).values
    assert float((near < 3).float().mean()) > 0.97


# ---- the multivariate Gaussian with a Wishart prior on the Cholesky factor of its precision
# (exploratory_experiments/multivariate-gaussian-test.ipynb cell 15) ----
def _mvn_wishart_model(dims=2, wishart_scale=None):
    from bayes_vi_amd import linalg as tfl
    priors = collections.OrderedDict(
        loc=tfd.MultivariateNormalDiag(loc=[0.] * dims, scale_diag=[10.] * dims),
        precision=tfd.WishartTriL(df=5, scale_tril=10.0 * torch.eye(dims) if wishart_scale is None else wishart_scale,
                                  input_output_cholesky=True))

    def likelihood(loc, precision):
        return tfd.MultivariateNormalTriL(loc=loc, scale_tril=tfl.cholesky_solve(precision, tfl.eye(dims)))

    bij = [tfb.Identity(), tfb.FillScaleTriL(diag_bijector=tfb.Softplus(), diag_shift=1e-3)]
    return Model(priors=priors, likelihood=likelihood, constraining_bijectors=bij)


def test_wishart_tril_against_scipy_and_oracle():
    from scipy import stats
    rng = np.random.default_rng(5)
    for d, df in ((2, 5.0), (3, 4.5), (5, 9.0)):
        S = np.tril(rng.normal(size=(d, d)) * 0.3) + np.eye(d) * 2.0
        L = np.tril(rng.normal(size=(6, d, d)) * 0.5)
        for i in range(d):
            L[:, i, i] = rng.uniform(0.5, 2.0, 6)
        dist = tfd.WishartTriL(df=df, scale_tril=torch.tensor(S), input_output_cholesky=True)
        x = torch.tensor(L.reshape(6, -1))
        lp, g = dist.log_prob_flat(x).numpy(), dist.score_flat(x).numpy()
        ref = np.array([stats.wishart.logpdf(l @ l.T, df, S @ S.T) for l in L])
        np.testing.assert_allclose(lp, ref, rtol=2e-6, atol=2e-5)            # hyper-parameters are held in float32
        olp, og = cf.wishart_tril_prior(L.reshape(6, -1), df, S)
        np.testing.assert_allclose(lp, olp, rtol=2e-6, atol=2e-5)
        np.testing.assert_allclose(g, og, rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(dist.log_prob(torch.tensor(L)).numpy(), lp)
    # draws are Cholesky factors with E[L L^T] = df V
    dist = tfd.WishartTriL(df=7.0, scale_tril=torch.tensor([[2.0, 0.0], [0.5, 1.0]]), input_output_cholesky=True)
    gen = torch.Generator().manual_seed(1)
    Ls = dist.sample((40000,), generator=gen)
    assert Ls.shape == (40000, 2, 2) and bool((Ls[:, 0, 1] == 0).all()) and bool((torch.diagonal(Ls, dim1=-2, dim2=-1) > 0).all())
    V = dist.scale_tril @ dist.scale_tril.T
    np.testing.assert_allclose((Ls @ Ls.transpose(-1, -2)).mean(0).numpy(), 7.0 * V.numpy(), rtol=0.03, atol=0.1)
    with pytest.raises(NotImplementedError):
        tfd.WishartTriL(df=5, scale_tril=torch.eye(2))


def test_recognise_mvn_with_cholesky_precision():
    from bayes_vi_amd import linalg as tfl
    from bayes_vi_amd import _native
    from bayes_vi_amd.fused_target import Posterior
    m = _mvn_wishart_model(2)
    assert m.param_names == ['loc', 'precision']
    assert m.block_sizes == [2, 4] and m.unconstrained_block_sizes == [2, 3]
    assert m.flat_unconstrained_param_event_ndims == 5
    assert m.constraining_bijectors[1].d == 2 and m.constraining_bijectors[1].diag_shift == 1e-3
    desc = m.describe_likelihood()
    assert desc == dict(family='mvn_tril', d=2, parameterisation='cholesky_solve_identity', index=[0, 1, 2, 3, 4, 5])
    # scale_tril given directly, parameters in the other order
    priors = collections.OrderedDict(s=tfd.WishartTriL(df=4, scale_tril=torch.eye(3), input_output_cholesky=True),
                                     mu=tfd.MultivariateNormalDiag(loc=[0.] * 3, scale_diag=[1., 2., 3.]))
    m2 = Model(priors=priors, likelihood=lambda s, mu: tfd.MultivariateNormalTriL(loc=mu, scale_tril=s))
    d2 = m2.describe_likelihood()
    assert d2['parameterisation'] == 'scale_tril' and d2['index'] == [9, 10, 11] + list(range(9))
    assert isinstance(m2.constraining_bijectors[0], tfb.FillScaleTriL) and m2.constraining_bijectors[0].d == 3
    with pytest.raises(sym.UnsupportedLikelihood):
        Model(priors=priors, likelihood=lambda s, mu: tfd.MultivariateNormalTriL(
            loc=mu, scale_tril=tfl.cholesky_solve(s, tfl.eye(2)))).describe_likelihood()
    # the device-side description of priors and bijectors: Normal(0, 10) on loc, Wishart(5, 10 I) on the factor
    class _Lik:
        n_params = 6
    post = Posterior(_Lik(), [(p, n) for p, n in zip(m.priors.values(), m.block_sizes)], m.constraining_bijectors)
    blocks = post.device_blocks()
    assert blocks is not None and len(blocks) == 2
    assert (blocks[0].bijector, blocks[0].prior, blocks[0].a, blocks[0].b) == (_native.BIJ_IDENTITY, _native.PRIOR_NORMAL, 0.0, 10.0)
    assert (blocks[1].bijector, blocks[1].prior, blocks[1].a, blocks[1].b, blocks[1].dim) == \
        (_native.BIJ_FILL_SCALE_TRIL, _native.PRIOR_WISHART_TRIL, 5.0, 10.0, 2)
    assert abs(blocks[1].shift - 1e-3) < 1e-9
    # a Wishart scale that is not a multiple of the identity, or per-element prior widths: the torch path
    m3 = _mvn_wishart_model(2, wishart_scale=torch.tensor([[10.0, 0.0], [1.0, 10.0]]))
    post3 = Posterior(_Lik(), [(p, n) for p, n in zip(m3.priors.values(), m3.block_sizes)], m3.constraining_bijectors)
    assert post3.device_blocks() is None
    # concrete tensors: cholesky_solve is the inverse of P P^T, whatever sits above the diagonal of P
    P = torch.tensor([[2.0, 7.0], [0.5, 1.5]])
    Pl = torch.tril(P)
    np.testing.assert_allclose(tfl.cholesky_solve(P, tfl.eye(2)).numpy(), torch.linalg.inv(Pl @ Pl.T).numpy(), rtol=1e-5)


def test_mvn_wishart_priors_and_bijector_match_oracle():
    m = _mvn_wishart_model(3)
    rng = np.random.default_rng(3)
    z = rng.normal(size=(4, m.flat_unconstrained_param_event_ndims))
    spec = cf.mvn_chol_precision_model(np.zeros((1, 3)))
    theta = m.blockwise_constraining_bijector.forward(torch.tensor(z)).numpy()
    np.testing.assert_allclose(theta, spec.constrain(z), rtol=1e-12, atol=1e-12)
    olp, og = spec.log_prior(theta)
    parts = torch.split(torch.tensor(theta), m.block_sizes, dim=-1)
    lp = sum(p.log_prob_flat(x) for p, x in zip(m.priors.values(), parts)).numpy()
    g = torch.cat([p.score_flat(x) for p, x in zip(m.priors.values(), parts)], -1).numpy()
    np.testing.assert_allclose(lp, olp, rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(g, og, rtol=1e-5, atol=1e-6)
