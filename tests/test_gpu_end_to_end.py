"""GPU: the drop-in surface end to end -- Model / MCMC.fit / VI.fit over the fused kernels --
against the fp64 oracle (posterior assembly) and against exact posterior moments (samplers)."""
import collections
import math

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import closed_form as cf
from parity_util import assert_grad_close, assert_logp_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def bvi(gpu_device):
    import bayes_vi_amd.distributions as tfd
    from bayes_vi_amd.inference.mcmc import MCMC
    from bayes_vi_amd.inference.mcmc.stepsize_adaptation_kernels import DualAveragingStepSizeAdaptation
    from bayes_vi_amd.inference.mcmc.transition_kernels import (HamiltonianMonteCarlo, NoUTurnSampler,
                                                                 RandomWalkMetropolis)
    from bayes_vi_amd.inference.vi import VI
    from bayes_vi_amd.inference.vi.surrogate_posteriors import ADVI, NormalizingFlow
    from bayes_vi_amd.inference.vi.flow_bijectors import HamiltonianFlow
    from bayes_vi_amd.inference.vi.vi import Adam
    from bayes_vi_amd.model import Model
    from bayes_vi_amd.utils.datasets import make_dataset_from_df
    from bayes_vi_amd import symbolic
    ns = dict(locals()); ns.pop('gpu_device')
    return collections.namedtuple('bvi', ns.keys())(**ns)


def normal_setup(bvi, y):
    priors = collections.OrderedDict(loc=bvi.tfd.Normal(loc=0., scale=10.), scale=bvi.tfd.HalfNormal(scale=10.))
    model = bvi.Model(priors=priors, likelihood=lambda loc, scale: bvi.tfd.Normal(loc=loc, scale=scale))
    dataset = bvi.make_dataset_from_df(pd.DataFrame(dict(y=y)), target_names=['y'], format_features_as='dict')
    return model, dataset


def exact_normal_moments(y):
    spec = cf.normal_model(y)
    m, s = y.mean(), y.std(ddof=1)
    L, S = np.meshgrid(np.linspace(m - 6 * s / math.sqrt(len(y)), m + 6 * s / math.sqrt(len(y)), 241),
                       np.linspace(s * 0.35, s * 3.0, 401), indexing='ij')
    lp, _ = spec.log_post_constrained(np.stack([L.ravel(), S.ravel()], -1))
    w = np.exp(lp - lp.max()).reshape(L.shape); w /= w.sum()
    ml, ms = (w * L).sum(), (w * S).sum()
    return ml, math.sqrt((w * (L - ml) ** 2).sum()), ms, math.sqrt((w * (S - ms) ** 2).sum())


def test_posterior_assembly_matches_oracle_all_families(bvi, gpu_device):
    rng = np.random.default_rng(0)
    # iid Normal (README.md:114-120)
    y = rng.normal(7, 3, 20).astype(np.float32)
    model, ds = normal_setup(bvi, y)
    post = bvi.MCMC(model, ds).posterior
    z = rng.uniform(-1, 1, (6, 2)) + np.array([6.0, 1.5])
    lp, g = post.unconstrained_log_prob_and_grad(torch.tensor(z, dtype=torch.float32, device=gpu_device))
    rlp, rg = cf.normal_model(y).log_post_unconstrained(z.astype(np.float32))
    assert_logp_close(lp.cpu().numpy(), rlp); assert_grad_close(g.cpu().numpy(), rg)
    # regression with dict features (README.md:745-756)
    x = rng.normal(0, 10, 50).astype(np.float32); yy = (7 + 3 * x + rng.normal(0, 10, 50)).astype(np.float32)
    priors = collections.OrderedDict(beta_0=bvi.tfd.Normal(0., 10.), beta_1=bvi.tfd.Normal(0., 10.),
                                     scale=bvi.tfd.HalfNormal(10.))

    def likelihood(beta_0, beta_1, scale, features):
        return bvi.tfd.Normal(loc=beta_0 + beta_1 * features['x'], scale=scale)
    model = bvi.Model(priors, likelihood)
    ds = bvi.make_dataset_from_df(pd.DataFrame(dict(x=x, y=yy)), target_names=['y'], feature_names=['x'])
    post = bvi.MCMC(model, ds).posterior
    z = rng.uniform(-1, 1, (5, 3)) + np.array([7.0, 3.0, 9.0])
    lp, g = post.unconstrained_log_prob_and_grad(torch.tensor(z, dtype=torch.float32, device=gpu_device))
    rlp, rg = cf.linreg_model(x[:, None], yy).log_post_unconstrained(z.astype(np.float32))
    assert_logp_close(lp.cpu().numpy(), rlp); assert_grad_close(g.cpu().numpy(), rg)
    # vector slopes via matvec, parameters declared in a different order than the kernel's
    d = 16
    X = rng.normal(size=(3000, d)).astype(np.float32); w = rng.normal(size=d)
    y3 = (0.5 + X @ w + rng.normal(size=3000)).astype(np.float32)
    priors = collections.OrderedDict(scale=bvi.tfd.HalfNormal(10.), w=bvi.tfd.Normal(torch.zeros(d), 10.),
                                     b=bvi.tfd.Normal(0., 10.))
    model = bvi.Model(priors, lambda scale, w, b, features: bvi.tfd.Normal(loc=bvi.symbolic.matvec(features, w) + b,
                                                                           scale=scale))
    df = pd.DataFrame(X, columns=['f%d' % i for i in range(d)]); df['y'] = y3
    ds = bvi.make_dataset_from_df(df, ['y'], ['f%d' % i for i in range(d)], format_features_as='tensor')
    post = bvi.MCMC(model, ds).posterior
    zc = np.concatenate([rng.uniform(0.3, 1.0, (4, 1)), w + 0.05 * rng.normal(size=(4, d)),
                         0.5 + 0.05 * rng.normal(size=(4, 1))], -1)
    lp, g = post.unconstrained_log_prob_and_grad(torch.tensor(zc, dtype=torch.float32, device=gpu_device))
    spec = cf.linreg_model(X, y3)            # oracle layout: (b, w, scale)
    zo = np.concatenate([zc[:, d + 1:], zc[:, 1:d + 1], zc[:, :1]], -1).astype(np.float32)
    rlp, rg = spec.log_post_unconstrained(zo)
    rg_model = np.concatenate([rg[:, d + 1:], rg[:, 1:d + 1], rg[:, :1]], -1)
    assert_logp_close(lp.cpu().numpy(), rlp); assert_grad_close(g.cpu().numpy(), rg_model)
    # univariate mixture (univariate-gaussian-mixture-test.ipynb cell 15)
    k = 3
    comp = rng.integers(0, k, 500)
    ym = rng.normal(np.array([-6.0, 0.0, 5.0])[comp], 1.0).astype(np.float32)
    priors = collections.OrderedDict(weights=bvi.tfd.Dirichlet(torch.ones(k)),
                                     loc=bvi.tfd.Normal(torch.zeros(k), torch.ones(k) * 10.),
                                     scale=bvi.tfd.HalfNormal(torch.ones(k) * 10.))
    model = bvi.Model(priors, lambda weights, loc, scale: bvi.tfd.MixtureSameFamily(
        mixture_distribution=bvi.tfd.Categorical(probs=weights),
        components_distribution=bvi.tfd.Normal(loc=loc, scale=scale)))
    ds = bvi.make_dataset_from_df(pd.DataFrame(dict(y=ym)), ['y'])
    post = bvi.MCMC(model, ds).posterior
    z = rng.uniform(-1, 1, (4, 8)) + np.array([0, 0, -6, 0, 5, 0.5, 0.5, 0.5])
    lp, g = post.unconstrained_log_prob_and_grad(torch.tensor(z, dtype=torch.float32, device=gpu_device))
    rlp, rg = cf.gmm1d_model(ym, k).log_post_unconstrained(z.astype(np.float32))
    assert_logp_close(lp.cpu().numpy(), rlp); assert_grad_close(g.cpu().numpy(), rg)


def test_config1_nuts_dual_averaging(bvi):
    """BASELINE.json configs[0]: univariate Gaussian, 20 points, NUTS + DualAveraging, 2 chains
    (README.md:146-162; draws reduced to keep the test short).  Moments within Monte-Carlo error of
    the exact posterior moments (quadrature of the oracle's log posterior)."""
    rng = np.random.default_rng(1)
    y = rng.normal(7, 3, 20).astype(np.float32)
    model, ds = normal_setup(bvi, y)
    kernel = bvi.NoUTurnSampler(stepsize_adaptation_kernel=bvi.DualAveragingStepSizeAdaptation(num_adaptation_steps=400))
    res = bvi.MCMC(model=model, dataset=ds, transition_kernel=kernel).fit(
        num_chains=2, num_samples=1500, num_burnin_steps=500, progress_bar=False, seed=5)
    loc, scale = (s.cpu().numpy() for s in res.samples)
    assert loc.shape == (1500, 2) and scale.shape == (1500, 2)
    ml, sl, ms, ss = exact_normal_moments(y.astype(np.float64))
    # ESS of NUTS here is ~ the number of draws; allow 5 standard errors of 3000 draws + 3 % of sd
    assert loc.mean() == pytest.approx(ml, abs=5 * sl / math.sqrt(3000) + 0.03 * sl)
    assert scale.mean() == pytest.approx(ms, abs=5 * ss / math.sqrt(3000) + 0.03 * ss)
    assert loc.std() == pytest.approx(sl, rel=0.12) and scale.std() == pytest.approx(ss, rel=0.15)
    assert set(res.trace.keys()) == {'is_accepted', 'target_log_prob', 'log_accept_ratio', 'step_size',
                                     'leapfrogs_taken', 'has_divergence', 'energy'}
    assert tuple(res.accept_ratios.shape) == (2,) and float(res.accept_ratios.min()) > 0.5
    assert int(res.trace['leapfrogs_taken'].max()) <= 31


def test_regression_rwm_and_hmc(bvi):
    """README.md:772-791 (RandomWalkMetropolis on the 1-feature regression) and HMC on the same model."""
    rng = np.random.default_rng(2)
    x = rng.normal(0, 10, 50).astype(np.float32); yy = (7 + 3 * x + rng.normal(0, 10, 50)).astype(np.float32)
    priors = collections.OrderedDict(beta_0=bvi.tfd.Normal(0., 10.), beta_1=bvi.tfd.Normal(0., 10.),
                                     scale=bvi.tfd.HalfNormal(10.))

    def likelihood(beta_0, beta_1, scale, features):
        return bvi.tfd.Normal(loc=beta_0 + beta_1 * features['x'], scale=scale)
    model = bvi.Model(priors, likelihood)
    ds = bvi.make_dataset_from_df(pd.DataFrame(dict(x=x, y=yy)), target_names=['y'], feature_names=['x'])
    A = np.stack([np.ones(50), x], -1); ols = np.linalg.lstsq(A, yy, rcond=None)[0]
    init = [torch.full((4,), 5.0), torch.full((4,), 2.0), torch.full((4,), 8.0)]
    hmc = bvi.HamiltonianMonteCarlo(step_size=0.05, num_leapfrog_steps=10,
                                    stepsize_adaptation_kernel=bvi.DualAveragingStepSizeAdaptation(300))
    res = bvi.MCMC(model, ds, hmc).fit(initial_state=init, num_samples=1500, num_burnin_steps=500,
                                       progress_bar=False, seed=1)
    b0, b1, sc = (s.cpu().numpy() for s in res.samples)
    assert b0.shape == (1500, 4)
    assert b1.mean() == pytest.approx(ols[1], abs=0.05)
    assert b0.mean() == pytest.approx(ols[0], abs=0.6)
    assert set(res.trace.keys()) == {'is_accepted', 'target_log_prob', 'step_size'}
    rwm = bvi.MCMC(model, ds, bvi.RandomWalkMetropolis()).fit(initial_state=init, num_samples=3000,
                                                              num_burnin_steps=1000, progress_bar=False, seed=2)
    assert set(rwm.trace.keys()) == {'is_accepted'}
    assert rwm.samples[1].cpu().numpy().mean() == pytest.approx(ols[1], abs=0.15)


def test_public_log_prob_stays_absolute_after_a_device_sampler_ran(bvi, gpu_device):
    """The device samplers work on log-densities relative to a level they keep on the device; `MCMC.target_log_prob`
    (the closure of mcmc.py:57-62), `Posterior.log_prob` and `unconstrained_log_prob_and_grad` on the same -- cached --
    posterior must return the same absolute values before and after such a run, for every device kernel."""
    rng = np.random.default_rng(21)
    n, d = 70_000, 8             # beyond the rows the whole-transition kernels take (those keep absolute floats)
    X = rng.normal(size=(n, d)).astype(np.float32); w = rng.normal(size=d)
    yy = (0.5 + X @ w + rng.normal(size=n)).astype(np.float32)
    df = pd.DataFrame({**{'f%d' % i: X[:, i] for i in range(d)}, 'y': yy})
    ds = bvi.make_dataset_from_df(df, ['y'], ['f%d' % i for i in range(d)], format_features_as='tensor')
    priors = collections.OrderedDict(b=bvi.tfd.Normal(0., 10.), w=bvi.tfd.Normal(torch.zeros(d), 10.),
                                     scale=bvi.tfd.HalfNormal(10.))
    model = bvi.Model(priors, lambda b, w, scale, features: bvi.tfd.Normal(
        loc=b + bvi.symbolic.matvec(features, w), scale=scale))
    state = [torch.tensor([0.4, 0.6]), torch.tensor(np.stack([w, w + 0.01]), dtype=torch.float32), torch.tensor([1.0, 1.1])]
    z = torch.tensor(rng.normal(0, 0.1, (3, d + 2)), dtype=torch.float32, device=gpu_device)
    spec = cf.linreg_model(X, yy)
    ref = spec.log_post_constrained(np.concatenate([s.reshape(2, -1).numpy() for s in state], -1))[0]
    assert abs(ref).min() > 5e4                     # a level that leaked would show as an error of this size
    for kernel in (bvi.HamiltonianMonteCarlo(step_size=1e-3, num_leapfrog_steps=3),
                   bvi.NoUTurnSampler(step_size=1e-3, max_tree_depth=3), bvi.RandomWalkMetropolis()):
        mcmc = bvi.MCMC(model, ds, kernel)
        before = mcmc.target_log_prob(state).cpu().numpy()
        before_u = mcmc.posterior.unconstrained_log_prob_and_grad(z)[0].cpu().numpy()
        assert_logp_close(before, ref)
        res = mcmc.fit(num_chains=4, num_samples=5, num_burnin_steps=5, progress_bar=False, seed=1)
        assert mcmc.posterior.likelihood.logp_ref is not None      # a device kernel ran and kept a level
        np.testing.assert_array_equal(mcmc.target_log_prob(state).cpu().numpy(), before)
        np.testing.assert_array_equal(mcmc.posterior.unconstrained_log_prob_and_grad(z)[0].cpu().numpy(), before_u)
        # and what the run itself reported is absolute as well
        tl = res.trace['target_log_prob'] if 'target_log_prob' in res.trace else None
        if tl is not None:
            assert float(tl.abs().min()) > 5e4


def test_vi_meanfield_advi_and_hnf(bvi):
    """README.md:264-279 (mean-field ADVI) and :626-637 (Hamiltonian normalizing flow)."""
    rng = np.random.default_rng(3)
    y = rng.normal(7, 3, 100).astype(np.float32)
    model, ds = normal_setup(bvi, y)
    ml, sl, ms, ss = exact_normal_moments(y.astype(np.float64))
    torch.manual_seed(0)
    vi = bvi.VI(model=model, dataset=ds, surrogate_posterior=bvi.ADVI(model=model, mean_field=True))
    approx, losses = vi.fit(optimizer=bvi.Adam(learning_rate=5e-2), num_steps=1500, sample_size=10,
                            num_samples_to_approx_marginals=4000, progress_bar=False)
    assert losses.shape == (1500,) and losses[-100:].mean() < losses[:50].mean()
    assert float(approx['loc'].mean()) == pytest.approx(ml, abs=4 * sl)
    assert float(approx['scale'].mean()) == pytest.approx(ms, abs=4 * ss)
    assert float(approx['loc'].stddev()) == pytest.approx(sl, rel=0.5)
    # ELBO <= log evidence: final loss above the negative log marginal likelihood (from quadrature)
    spec = cf.normal_model(y.astype(np.float64))
    L, S = np.meshgrid(np.linspace(ml - 8 * sl, ml + 8 * sl, 321), np.linspace(ms * 0.4, ms * 2.5, 401), indexing='ij')
    lp, _ = spec.log_post_constrained(np.stack([L.ravel(), S.ravel()], -1))
    log_ev = np.log(np.exp(lp - lp.max()).sum() * (L[1, 0] - L[0, 0]) * (S[0, 1] - S[0, 0])) + lp.max()
    assert losses[-200:].mean() > -log_ev - 0.05
    assert losses[-200:].mean() < -log_ev + 1.0
    # full-rank ADVI and an augmented Hamiltonian flow run and improve the loss
    vi = bvi.VI(model=model, dataset=ds, surrogate_posterior=bvi.ADVI(model=model, mean_field=False))
    _, l2 = vi.fit(optimizer=bvi.Adam(learning_rate=5e-2), num_steps=300, sample_size=10,
                   num_samples_to_approx_marginals=100, progress_bar=False)
    assert l2[-50:].mean() < l2[:20].mean()
    nd = model.flat_unconstrained_param_event_ndims
    flow = bvi.HamiltonianFlow(nd, step_sizes=torch.nn.Parameter(torch.tensor(0.1)), num_integration_steps=2)
    sp = bvi.NormalizingFlow(model=model, flow_bijector=flow, extra_ndims=nd)
    vi = bvi.VI(model=model, dataset=ds, surrogate_posterior=sp)
    _, l3 = vi.fit(optimizer=bvi.Adam(learning_rate=1e-2), num_steps=300, sample_size=10,
                   num_samples_to_approx_marginals=100, progress_bar=False)
    assert np.isfinite(l3).all() and l3[-50:].mean() < l3[:20].mean()


def test_vi_graph_capture_agrees_with_eager_steps(bvi):
    """VI.fit replays the whole optimisation step as a captured HIP graph by default; it must optimise
    the same objective as step-by-step execution: same converged loss and fitted posterior moments."""
    rng = np.random.default_rng(3)
    y = rng.normal(7, 3, 100).astype(np.float32)
    model, ds = normal_setup(bvi, y)
    out = {}
    for mode in (False, True):
        torch.manual_seed(0)
        vi = bvi.VI(model=model, dataset=ds, surrogate_posterior=bvi.ADVI(model, mean_field=True))
        post, losses = vi.fit(optimizer=bvi.Adam(learning_rate=5e-2), num_steps=1500, sample_size=20,
                              num_samples_to_approx_marginals=4000, progress_bar=False, capture_graph=mode)
        assert losses.shape == (1500,) and np.isfinite(losses).all()
        out[mode] = (losses[-300:].mean(), post)
    assert out[True][0] == pytest.approx(out[False][0], rel=5e-3)
    ml, sl, ms, ss = exact_normal_moments(y.astype(np.float64))
    for mode in (False, True):
        assert float(out[mode][1]['loc'].mean()) == pytest.approx(ml, abs=0.15)
        assert float(out[mode][1]['scale'].mean()) == pytest.approx(ms, abs=0.15)


def test_device_philox_matches_oracle(gpu_device):
    from bayes_vi_amd import _native
    from oracle import philox
    raw = _native.philox_uint32(0x1234567890ABCDEF, (5 << 32) | 77, 1, 9, 5, gpu_device).cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(raw, philox.draw_blocks(0x1234567890ABCDEF, (5 << 32) | 77, 1, 9, 5))
    z = _native.philox_normal(42, 3, 50, 23, gpu_device).cpu().numpy()
    np.testing.assert_allclose(z, philox.normals(42, 3, 50, 23), rtol=2e-5, atol=2e-6)


def test_device_hmc_matches_oracle_with_injected_and_philox_randomness(bvi, gpu_device):
    """bvi_hmc_begin / bvi_leapfrog_pre / bvi_leapfrog_post / bvi_hmc_end against the oracle's HMC
    step, first with injected momenta and uniforms, then with the Philox streams both sides draw."""
    from bayes_vi_amd.inference.mcmc.device_hmc import DeviceHamiltonianMonteCarloKernel
    from oracle import leapfrog as lf, philox
    rng = np.random.default_rng(4)
    k = 3
    comp = rng.integers(0, k, 400)
    ym = rng.normal(np.array([-6.0, 0.0, 5.0])[comp], 1.0).astype(np.float32)
    priors = collections.OrderedDict(weights=bvi.tfd.Dirichlet(torch.ones(k)),
                                     loc=bvi.tfd.Normal(torch.zeros(k), torch.ones(k) * 10.),
                                     scale=bvi.tfd.HalfNormal(torch.ones(k) * 10.))
    model = bvi.Model(priors, lambda weights, loc, scale: bvi.tfd.MixtureSameFamily(
        mixture_distribution=bvi.tfd.Categorical(probs=weights),
        components_distribution=bvi.tfd.Normal(loc=loc, scale=scale)))
    ds = bvi.make_dataset_from_df(pd.DataFrame(dict(y=ym)), ['y'])
    post = bvi.MCMC(model, ds).posterior
    assert post.device_blocks() is not None
    spec = cf.gmm1d_model(ym, k)
    c = 7
    z = (rng.uniform(-.3, .3, (c, 8)) + np.array([0, 0, -6, 0, 5, 0.5, 0.5, 0.5])).astype(np.float32)
    p = rng.normal(size=(c, 8)).astype(np.float32); lu = np.log(rng.uniform(size=c)).astype(np.float32)
    kern = DeviceHamiltonianMonteCarloKernel(post, step_size=0.01, num_leapfrog_steps=4, seed=99)
    zt = torch.tensor(z, device=gpu_device)
    st = kern.bootstrap(zt)
    rlp, rg = spec.log_post_unconstrained(z)
    assert_logp_close(st['target_log_prob'].cpu().numpy(), rlp); assert_grad_close(st['grad'].cpu().numpy(), rg)
    z1, st1, tr = kern.one_step(zt, st, momenta=torch.tensor(p, device=gpu_device),
                                log_uniform=torch.tensor(lu, device=gpu_device))
    q_ref, acc_ref, lr_ref, lp_ref = lf.hmc_step(spec.log_post_unconstrained, z, p, lu, 0.01, 4)
    np.testing.assert_array_equal(tr['is_accepted'].cpu().numpy(), acc_ref)
    np.testing.assert_allclose(z1.cpu().numpy(), q_ref, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(tr['log_accept_ratio'].cpu().numpy(), lr_ref, atol=5e-3)
    assert_logp_close(st1['target_log_prob'].cpu().numpy(), lp_ref, rtol=2e-5)
    assert torch.equal(zt, torch.tensor(z, device=gpu_device))       # the caller's state is not modified
    # Philox path: the oracle draws the same momenta / uniforms from its numpy generator
    z2, st2, tr2 = kern.one_step(z1, st1)
    pm = philox.normals(99, 1, c, 8); lu2 = philox.log_uniform(99, 1, c)
    q_ref2, acc_ref2, lr_ref2, _ = lf.hmc_step(spec.log_post_unconstrained, z1.cpu().numpy(), pm, lu2, 0.01, 4)
    safe = np.abs(lu2 - lr_ref2) > 1e-2                              # chains not sitting on the accept boundary
    np.testing.assert_array_equal(tr2['is_accepted'].cpu().numpy()[safe], acc_ref2[safe])
    np.testing.assert_allclose(z2.cpu().numpy()[safe], q_ref2[safe], rtol=5e-5, atol=5e-5)


def test_device_hmc_graph_replay_gives_the_chain_of_single_steps(bvi, gpu_device):
    """DeviceHamiltonianMonteCarloKernel.many_steps (one transition recorded as a HIP graph, Philox
    transition number advanced on the device) against the same kernel stepped launch by launch: same
    positions, accept decisions and log densities, transition for transition."""
    from bayes_vi_amd.inference.mcmc.device_hmc import DeviceHamiltonianMonteCarloKernel
    rng = np.random.default_rng(77)
    for name, model, ds, spec, z in _fused_cases(bvi, rng):
        post = bvi.MCMC(model, ds).posterior
        eps = {'normal': 0.2, 'linreg': 0.01, 'gmm1d': 0.04}[name]
        zt = torch.tensor(z.astype(np.float32), device=gpu_device)
        T = 12
        a = DeviceHamiltonianMonteCarloKernel(post, step_size=eps, num_leapfrog_steps=3, seed=21, graph_replay=True)
        zs, zf, stf, tr = a.many_steps(zt, a.bootstrap(zt), T)
        assert a.transition == T
        # its own posterior: the level the stored log-densities are relative to lives in the posterior's likelihood
        # and moves with the chains, and bitwise equality needs both kernels to start from the same one
        b = DeviceHamiltonianMonteCarloKernel(bvi.MCMC(model, ds).posterior, step_size=eps, num_leapfrog_steps=3, seed=21)
        zc, st = zt, b.bootstrap(zt)
        for t in range(T):
            zc, st, tr1 = b.one_step(zc, st)
            assert torch.equal(tr['is_accepted'][t], tr1['is_accepted']), (name, t)
            assert torch.equal(zs[t], zc), (name, t)
            assert torch.equal(tr['target_log_prob'][t], st['target_log_prob']), (name, t)
            assert torch.equal(tr['log_accept_ratio'][t], tr1['log_accept_ratio']), (name, t)
        assert torch.equal(zf, zc) and torch.equal(stf['grad'], st['grad'])
        assert 0 < float(tr['is_accepted'].float().mean()) <= 1
        # a second call continues the same chain
        zs2, _, _, _ = a.many_steps(zf, stf, 9)
        zc2, st2, _ = b.one_step(zc, st)
        assert torch.equal(zs2[0], zc2), name


def _fused_cases(bvi, rng):
    """(name, model, dataset, oracle spec, start point z[C, Pu]) for the three families of the fused HMC kernel."""
    out = []
    y = rng.normal(7, 3, 20).astype(np.float32)
    model, ds = normal_setup(bvi, y)
    out.append(('normal', model, ds, cf.normal_model(y), rng.uniform(-.5, .5, (5, 2)) + np.array([6.0, 1.5])))
    d = 3
    X = rng.normal(size=(1500, d)).astype(np.float32); w = rng.normal(size=d)
    yr = (0.5 + X @ w + rng.normal(size=1500)).astype(np.float32)
    priors = collections.OrderedDict(b=bvi.tfd.Normal(0., 10.), w=bvi.tfd.Normal(torch.zeros(d), 10.),
                                     scale=bvi.tfd.HalfNormal(10.))
    model = bvi.Model(priors, lambda b, w, scale, features: bvi.tfd.Normal(loc=b + bvi.symbolic.matvec(features, w),
                                                                           scale=scale))
    df = pd.DataFrame(X, columns=['f%d' % i for i in range(d)]); df['y'] = yr
    ds = bvi.make_dataset_from_df(df, ['y'], ['f%d' % i for i in range(d)], format_features_as='tensor')
    z0 = np.concatenate([0.5 + 0.05 * rng.normal(size=(6, 1)), w + 0.05 * rng.normal(size=(6, d)),
                         0.5 + 0.05 * rng.normal(size=(6, 1))], -1)
    out.append(('linreg', model, ds, cf.linreg_model(X, yr), z0))
    k = 3
    comp = rng.integers(0, k, 400)
    ym = rng.normal(np.array([-6.0, 0.0, 5.0])[comp], 1.0).astype(np.float32)
    priors = collections.OrderedDict(weights=bvi.tfd.Dirichlet(torch.ones(k)),
                                     loc=bvi.tfd.Normal(torch.zeros(k), torch.ones(k) * 10.),
                                     scale=bvi.tfd.HalfNormal(torch.ones(k) * 10.))
    model = bvi.Model(priors, lambda weights, loc, scale: bvi.tfd.MixtureSameFamily(
        mixture_distribution=bvi.tfd.Categorical(probs=weights),
        components_distribution=bvi.tfd.Normal(loc=loc, scale=scale)))
    ds = bvi.make_dataset_from_df(pd.DataFrame(dict(y=ym)), ['y'])
    out.append(('gmm1d', model, ds, cf.gmm1d_model(ym, k),
                rng.uniform(-.3, .3, (7, 8)) + np.array([0, 0, -6, 0, 5, 0.5, 0.5, 0.5])))
    return out


def test_fused_hmc_kernel_matches_oracle_and_unfused_device_path(bvi, gpu_device):
    """bvi_hmc_fused (whole transitions in one launch): the evaluation at the start point and one
    transition with injected randomness against the oracle; five Philox-driven transitions in ONE
    launch against five transitions of the launch-per-step device path with the same seed."""
    from bayes_vi_amd.inference.mcmc.device_hmc import (DeviceHamiltonianMonteCarloKernel,
                                                        FusedHamiltonianMonteCarloKernel, fused_family)
    from oracle import leapfrog as lf
    rng = np.random.default_rng(21)
    for name, model, ds, spec, z in _fused_cases(bvi, rng):
        post = bvi.MCMC(model, ds).posterior
        fam = fused_family(post)
        assert fam is not None, name
        z = z.astype(np.float32)
        c, pu = z.shape
        eps, nl = (0.05, 4) if name == 'normal' else (0.004, 4)
        fused = FusedHamiltonianMonteCarloKernel(post, fam, step_size=eps, num_leapfrog_steps=nl, seed=7)
        zt = torch.tensor(z, device=gpu_device)
        st = fused.bootstrap(zt)
        rlp, rg = spec.log_post_unconstrained(z)
        assert_logp_close(st['target_log_prob'].cpu().numpy(), rlp)
        assert_grad_close(st['grad'].cpu().numpy(), rg)
        p = rng.normal(size=(c, pu)).astype(np.float32); lu = np.log(rng.uniform(size=c)).astype(np.float32)
        z1, st1, tr = fused.one_step(zt, st, momenta=torch.tensor(p, device=gpu_device),
                                     log_uniform=torch.tensor(lu, device=gpu_device))
        q_ref, acc_ref, lr_ref, lp_ref = lf.hmc_step(spec.log_post_unconstrained, z, p, lu, eps, nl)
        safe = np.abs(lu - lr_ref) > 1e-2
        np.testing.assert_array_equal(tr['is_accepted'].cpu().numpy()[safe], acc_ref[safe])
        np.testing.assert_allclose(z1.cpu().numpy()[safe], q_ref[safe], rtol=3e-5, atol=3e-5)
        np.testing.assert_allclose(tr['log_accept_ratio'].cpu().numpy(), lr_ref, atol=5e-3)
        assert_logp_close(st1['target_log_prob'].cpu().numpy()[safe], lp_ref[safe], rtol=2e-5)
        assert torch.equal(zt, torch.tensor(z, device=gpu_device))
        # same seed, same transition counters: the fused launch and the per-step launches agree
        fused = FusedHamiltonianMonteCarloKernel(post, fam, step_size=eps, num_leapfrog_steps=nl, seed=7)
        step = DeviceHamiltonianMonteCarloKernel(post, step_size=eps, num_leapfrog_steps=nl, seed=7)
        zs, zf, stf, trf = fused.many_steps(zt, st, 5)
        zz, ss = zt, step.bootstrap(zt)
        ok = np.ones(c, bool)
        for t in range(5):
            zz, ss, trs = step.one_step(zz, ss)
            # a chain whose accept decision sits on the boundary may legitimately differ from here on
            ok &= np.abs(trs['log_accept_ratio'].cpu().numpy()
                         - trf['log_accept_ratio'][t].cpu().numpy()) < 1e-3
            ok &= trs['is_accepted'].cpu().numpy() == trf['is_accepted'][t].cpu().numpy()
            np.testing.assert_allclose(zs[t].cpu().numpy()[ok], zz.cpu().numpy()[ok], rtol=1e-4, atol=1e-4)
        assert ok.sum() >= c - 1, name
        assert torch.equal(zs[-1], zf)
        assert_logp_close(stf['target_log_prob'].cpu().numpy()[ok], ss['target_log_prob'].cpu().numpy()[ok], rtol=5e-5)


def test_fused_hmc_through_mcmc_fit_posterior_moments(bvi):
    """BASELINE.json configs[0] data (20 points) through MCMC.fit with HMC: the driver picks the fused
    kernel, adapts the step size transition by transition, then draws every sample in one launch.
    Posterior moments within Monte-Carlo error of the exact ones."""
    from bayes_vi_amd.inference.mcmc.device_hmc import FusedHamiltonianMonteCarloKernel
    rng = np.random.default_rng(1)
    y = rng.normal(7, 3, 20).astype(np.float32)
    model, ds = normal_setup(bvi, y)
    hmc = bvi.HamiltonianMonteCarlo(step_size=0.1, num_leapfrog_steps=8,
                                    stepsize_adaptation_kernel=bvi.DualAveragingStepSizeAdaptation(300))
    mcmc = bvi.MCMC(model=model, dataset=ds, transition_kernel=hmc)
    kernel, _, _ = hmc(mcmc.posterior.unconstrained_log_prob_and_grad)
    assert isinstance(kernel.inner, FusedHamiltonianMonteCarloKernel)
    res = mcmc.fit(num_chains=8, num_samples=4000, num_burnin_steps=500, progress_bar=False, seed=3)
    loc, scale = (s.cpu().numpy() for s in res.samples)
    assert loc.shape == (4000, 8) and res.trace['is_accepted'].shape == (4000, 8)
    assert 0.5 < float(res.trace['is_accepted'].float().mean()) < 0.98
    ml, sl, ms, ss = exact_normal_moments(y.astype(np.float64))
    n_eff = 4000 * 8 / 4.0      # HMC with 8 leapfrog steps decorrelates within a few transitions
    assert loc.mean() == pytest.approx(ml, abs=5 * sl / math.sqrt(n_eff) + 0.02 * sl)
    assert scale.mean() == pytest.approx(ms, abs=5 * ss / math.sqrt(n_eff) + 0.02 * ss)
    assert loc.std() == pytest.approx(sl, rel=0.08) and scale.std() == pytest.approx(ss, rel=0.10)


def test_fused_rwm_kernel_follows_oracle(bvi, gpu_device):
    """bvi_rwm_fused: 25 transitions in one launch against the Metropolis rule applied in float64 to the
    same Philox proposals and uniforms."""
    from bayes_vi_amd.inference.mcmc.device_hmc import FusedRandomWalkMetropolisKernel, fused_family
    from oracle import philox
    rng = np.random.default_rng(41)
    for name, model, ds, spec, z in _fused_cases(bvi, rng):
        post = bvi.MCMC(model, ds).posterior
        z = z.astype(np.float32)
        c, pu = z.shape
        scale = {'normal': 0.3, 'linreg': 0.01, 'gmm1d': 0.03}[name]
        kern = FusedRandomWalkMetropolisKernel(post, fused_family(post), scale=scale, seed=13)
        zt = torch.tensor(z, device=gpu_device)
        zs, zf, stf, tr = kern.many_steps(zt, kern.bootstrap(zt), 25)
        zs = zs.cpu().numpy(); acc = tr['is_accepted'].cpu().numpy(); lrs = tr['log_accept_ratio'].cpu().numpy()
        assert 0 < acc.mean() < 1, name
        prev = z.astype(np.float64)
        checked = 0
        for t in range(25):
            prop = prev + scale * philox.normals(13, t, c, pu).astype(np.float64)
            lr = spec.log_post_unconstrained(prop)[0] - spec.log_post_unconstrained(prev)[0]
            lu = philox.log_uniform(13, t, c)
            safe = np.abs(lu - lr) > 1e-2
            np.testing.assert_array_equal(acc[t][safe], (lu < lr)[safe])
            np.testing.assert_allclose(lrs[t], lr, atol=5e-3, rtol=1e-4)
            want = np.where((lu < lr)[:, None], prop, prev)
            np.testing.assert_allclose(zs[t][safe], want[safe], rtol=2e-5, atol=2e-5)
            checked += int(safe.sum())
            prev = zs[t].astype(np.float64)
        assert checked > 0.8 * 25 * c


def test_device_rwm_kernel_gives_the_chain_of_the_fused_kernel(bvi, gpu_device):
    """DeviceRandomWalkMetropolisKernel (begin / pre with zero gradient / value-only pass / assemble / end)
    draws the same Philox proposals and uniforms as bvi_rwm_fused, which is checked against the oracle
    above: both must make the same decisions and visit the same positions."""
    from bayes_vi_amd.inference.mcmc.device_hmc import (DeviceRandomWalkMetropolisKernel,
                                                        FusedRandomWalkMetropolisKernel, fused_family)
    rng = np.random.default_rng(43)
    for name, model, ds, spec, z in _fused_cases(bvi, rng):
        post = bvi.MCMC(model, ds).posterior
        scale = {'normal': 0.3, 'linreg': 0.01, 'gmm1d': 0.03}[name]
        zt = torch.tensor(z.astype(np.float32), device=gpu_device)
        fused = FusedRandomWalkMetropolisKernel(post, fused_family(post), scale=scale, seed=13)
        zs_f, _, _, tr_f = fused.many_steps(zt, fused.bootstrap(zt), 25)
        dev_k = DeviceRandomWalkMetropolisKernel(post, scale=scale, seed=13)
        zs_d, zf, stf, tr_d = dev_k.many_steps(zt, dev_k.bootstrap(zt), 25)
        lr_f, lr_d = tr_f['log_accept_ratio'].cpu().numpy(), tr_d['log_accept_ratio'].cpu().numpy()
        acc_f, acc_d = tr_f['is_accepted'].cpu().numpy(), tr_d['is_accepted'].cpu().numpy()
        # a decision can only differ where the float32 log ratios of the two evaluation orders straddle log u;
        # after such a split the chains differ, so compare chain by chain up to the first disagreement
        agree = 0
        for ch in range(z.shape[0]):
            for t in range(25):
                if acc_f[t, ch] != acc_d[t, ch]:
                    assert abs(lr_f[t, ch] - lr_d[t, ch]) < 5e-3, (name, ch, t)
                    break
                np.testing.assert_allclose(zs_d[t, ch].cpu().numpy(), zs_f[t, ch].cpu().numpy(), rtol=1e-5, atol=1e-5)
                assert abs(lr_f[t, ch] - lr_d[t, ch]) < 5e-3 + 1e-4 * abs(lr_f[t, ch]), (name, ch, t)
                agree += 1
        assert agree > 0.8 * 25 * z.shape[0], (name, agree)
        assert 0 < acc_d.mean() < 1
        z1, st1, tr1 = dev_k.one_step(zf, stf)      # continues the chain; the caller's tensors are untouched
        assert z1.shape == zf.shape and tr1['is_accepted'].dtype == torch.bool


def test_fused_nuts_kernel_follows_oracle_transition_by_transition(bvi, gpu_device):
    """bvi_nuts_fused against oracle/nuts.py, which draws the same Philox numbers: every transition of
    a 40-transition launch is replayed by the oracle from the device's own previous state and must give
    the same tree (leapfrog count), the same move decision and the same new position -- unless one of
    the oracle's comparisons was too close to call in float32 (its `margin`), which is rare."""
    from bayes_vi_amd.inference.mcmc.device_hmc import FusedNoUTurnSamplerKernel, fused_family
    from oracle import nuts
    rng = np.random.default_rng(31)
    for name, model, ds, spec, z in _fused_cases(bvi, rng):
        post = bvi.MCMC(model, ds).posterior
        fam = fused_family(post)
        z = z.astype(np.float32)[:3]
        c, pu = z.shape
        eps = {'normal': 0.2, 'linreg': 0.01, 'gmm1d': 0.04}[name]
        kern = FusedNoUTurnSamplerKernel(post, fam, step_size=eps, max_tree_depth=4, seed=5)
        zt = torch.tensor(z, device=gpu_device)
        st = kern.bootstrap(zt)
        T = 40
        zs, zf, stf, tr = kern.many_steps(zt, st, T)
        zs = zs.cpu().numpy(); lps = tr['target_log_prob'].cpu().numpy()
        leaps = tr['leapfrogs_taken'].cpu().numpy(); moved = tr['is_accepted'].cpu().numpy()
        energies = tr['energy'].cpu().numpy()
        lrs = tr['log_accept_ratio'].cpu().numpy()
        assert not tr['has_divergence'].any()
        assert leaps.min() >= 1 and leaps.max() <= 15, name
        f = lambda q: spec.log_post_unconstrained(q)
        compared = 0
        for ch in range(c):
            prev = z[ch].astype(np.float64)
            for t in range(T):
                lp0, g0 = f(prev[None])
                out = nuts.nuts_step(f, prev, float(lp0[0]), g0[0], chain=ch, seed=5, step=t, step_size=eps,
                                     max_tree_depth=4)
                if out['margin'] > 2e-3:
                    assert out['leapfrogs'] == leaps[t, ch], (name, ch, t)
                    assert out['moved'] == bool(moved[t, ch]), (name, ch, t)
                    np.testing.assert_allclose(zs[t, ch], out['z'], rtol=2e-4, atol=2e-4)
                    assert abs(lrs[t, ch] - out['log_accept_ratio']) < 5e-3
                    # log density minus kinetic energy at the selected point (tfp's kernel_results.energy)
                    assert abs(energies[t, ch] - out['energy']) <= 2e-5 * abs(out['energy']) + 5e-3, (name, ch, t)
                    compared += 1
                prev = zs[t, ch].astype(np.float64)
        assert compared > 0.6 * c * T, (name, compared)


def test_batched_device_nuts_follows_oracle_transition_by_transition(bvi, gpu_device):
    """The NUTS built from bvi_nuts_begin / subtree_begin / leaf / subtree_end + the chain-wise leapfrog
    halves + the batched likelihood kernels, against oracle/nuts.py with the same Philox numbers: chains
    advance in lock-step with frozen chains masked on the device, and every chain must still reproduce
    its own tree, move decision and new position."""
    from bayes_vi_amd.inference.mcmc.device_hmc import BatchedDeviceNoUTurnSamplerKernel
    from oracle import nuts
    rng = np.random.default_rng(32)
    for name, model, ds, spec, z in _fused_cases(bvi, rng):
        post = bvi.MCMC(model, ds).posterior
        z = z.astype(np.float32)[:5]
        c, pu = z.shape
        eps = {'normal': 0.2, 'linreg': 0.01, 'gmm1d': 0.04}[name]
        kern = BatchedDeviceNoUTurnSamplerKernel(post, step_size=eps, max_tree_depth=4, seed=9)
        zt = torch.tensor(z, device=gpu_device)
        st = kern.bootstrap(zt)
        T = 25
        f = lambda q: spec.log_post_unconstrained(q)
        compared, sizes = 0, set()
        for t in range(T):
            prev = zt.cpu().numpy().astype(np.float64)
            zt, st, tr = kern.one_step(zt, st)
            znew = zt.cpu().numpy()
            leaps = tr['leapfrogs_taken'].cpu().numpy(); moved = tr['is_accepted'].cpu().numpy()
            energies = tr['energy'].cpu().numpy()
            lrs = tr['log_accept_ratio'].cpu().numpy()
            assert not tr['has_divergence'].any()
            lp_new, g_new = f(znew.astype(np.float64))
            # the cached log density / gradient travel with the position
            np.testing.assert_allclose(st['target_log_prob'].cpu().numpy(), lp_new, rtol=2e-5, atol=2e-3)
            for ch in range(c):
                lp0, g0 = f(prev[ch][None])
                out = nuts.nuts_step(f, prev[ch], float(lp0[0]), g0[0], chain=ch, seed=9, step=t, step_size=eps,
                                     max_tree_depth=4)
                if out['margin'] > 2e-3:
                    assert out['leapfrogs'] == leaps[ch], (name, ch, t)
                    assert out['moved'] == bool(moved[ch]), (name, ch, t)
                    np.testing.assert_allclose(znew[ch], out['z'], rtol=2e-4, atol=2e-4)
                    assert abs(lrs[ch] - out['log_accept_ratio']) < 5e-3
                    assert abs(energies[ch] - out['energy']) <= 2e-5 * abs(out['energy']) + 5e-3, (name, ch, t)
                    compared += 1
                    sizes.add(int(leaps[ch]))
        assert compared > 0.6 * c * T, (name, compared)
        assert len(sizes) >= 2, (name, sizes)      # chains did stop at different tree sizes


def test_device_nuts_kernels_survive_divergent_trajectories(bvi, gpu_device):
    """A step size far too large: trajectories blow up (energy error beyond max_energy_diff, non-finite
    densities).  Both NUTS forms must flag the divergence, keep every chain at a finite state whose cached
    log density is the density of that state, and go on sampling afterwards."""
    from bayes_vi_amd.inference.mcmc.device_hmc import (BatchedDeviceNoUTurnSamplerKernel,
                                                        FusedNoUTurnSamplerKernel, fused_family)
    rng = np.random.default_rng(5)
    y = rng.normal(7, 3, 20).astype(np.float32)
    model, ds = normal_setup(bvi, y)
    post = bvi.MCMC(model, ds).posterior
    spec = cf.normal_model(y)
    z0 = torch.tensor(rng.uniform(-.5, .5, (6, 2)) + np.array([6.0, 1.5]), dtype=torch.float32, device=gpu_device)
    for kern in (BatchedDeviceNoUTurnSamplerKernel(post, step_size=40.0, max_tree_depth=5, seed=3),
                 FusedNoUTurnSamplerKernel(post, fused_family(post), step_size=40.0, max_tree_depth=5, seed=3)):
        z, st = z0, kern.bootstrap(z0)
        diverged = 0
        for t in range(12):
            z, st, tr = kern.one_step(z, st)
            diverged += int(tr['has_divergence'].sum())
            assert bool(torch.isfinite(z).all()) and bool(torch.isfinite(st['target_log_prob']).all())
            ref_lp, _ = spec.log_post_unconstrained(z.cpu().numpy().astype(np.float64))
            np.testing.assert_allclose(st['target_log_prob'].cpu().numpy(), ref_lp, rtol=2e-5, atol=2e-3)
            assert int(tr['leapfrogs_taken'].min()) >= 1
        assert diverged > 0, type(kern).__name__
        kern.step_size = 0.2                       # back to a sane step: the chains move again
        moved = 0
        for t in range(10):
            z, st, tr = kern.one_step(z, st)
            moved += int(tr['is_accepted'].sum())
        assert moved > 20 and not bool(tr['has_divergence'].any())


def test_batched_device_nuts_through_mcmc_fit_on_a_large_regression(bvi, gpu_device):
    """A regression too large for the one-workgroup-per-chain kernel: MCMC.fit selects the batched device
    NUTS; posterior moments against the exact conjugate answer (flat-ish priors, known noise scale is
    not assumed: compare the weights with ordinary least squares, whose sampling sd is sigma/sqrt(N))."""
    from bayes_vi_amd.inference.mcmc.device_hmc import BatchedDeviceNoUTurnSamplerKernel
    rng = np.random.default_rng(8)
    n, d = 100000, 3
    X = rng.normal(size=(n, d)).astype(np.float32)
    w_true, b_true, sig = np.array([0.7, -1.2, 0.4]), 0.3, 0.8
    y = (X @ w_true + b_true + sig * rng.normal(size=n)).astype(np.float32)
    priors = collections.OrderedDict(b=bvi.tfd.Normal(0., 10.), w=bvi.tfd.Normal(torch.zeros(d), 10.),
                                     scale=bvi.tfd.HalfNormal(10.))
    model = bvi.Model(priors, lambda b, w, scale, features: bvi.tfd.Normal(loc=b + bvi.symbolic.matvec(features, w),
                                                                           scale=scale))
    df = pd.DataFrame(X, columns=['f%d' % i for i in range(d)]); df['y'] = y
    ds = bvi.make_dataset_from_df(df, ['y'], ['f%d' % i for i in range(d)], format_features_as='tensor')
    cfg = bvi.NoUTurnSampler(step_size=0.002, max_tree_depth=5,
                             stepsize_adaptation_kernel=bvi.DualAveragingStepSizeAdaptation(num_adaptation_steps=150))
    mcmc = bvi.MCMC(model=model, dataset=ds, transition_kernel=cfg)
    kernel, trace_fns, names = cfg(mcmc.posterior.unconstrained_log_prob_and_grad)
    assert isinstance(kernel.inner, BatchedDeviceNoUTurnSamplerKernel)
    assert len(trace_fns) == 2 and all(callable(f) for f in trace_fns)      # (kernel, trace_fns, trace_metrics), transition_kernels.py:24-38
    from bayes_vi_amd.inference.mcmc.device_hmc import DeviceRandomWalkMetropolisKernel
    rwm = bvi.RandomWalkMetropolis()(mcmc.posterior.unconstrained_log_prob_and_grad)[0]
    assert isinstance(getattr(rwm, 'inner', rwm), DeviceRandomWalkMetropolisKernel)
    A = np.concatenate([X.astype(np.float64), np.ones((n, 1))], 1)
    ols = np.linalg.lstsq(A, y.astype(np.float64), rcond=None)[0]
    init = [torch.full((4,), float(ols[d])), torch.tensor(np.tile(ols[:d], (4, 1)), dtype=torch.float32),
            torch.full((4,), 1.0)]
    res = mcmc.fit(initial_state=init, num_samples=300, num_burnin_steps=200, progress_bar=False, seed=2)
    ws = res.samples[1].cpu().numpy().reshape(-1, d)
    sd = sig / math.sqrt(n)
    assert np.abs(ws.mean(0) - ols[:d]).max() < 6 * sd / math.sqrt(300 * 4 / 2) + 0.3 * sd
    assert ws.std(0) == pytest.approx(np.full(d, sd), rel=0.25)
    scale = res.samples[-1].cpu().numpy()
    assert scale.mean() == pytest.approx(sig, rel=0.02)
    assert not bool(res.trace['has_divergence'].any())


def test_fused_nuts_through_mcmc_fit(bvi):
    """BASELINE.json configs[0] through MCMC.fit: the driver selects the fused NUTS kernel; dual averaging runs
    transition by transition during adaptation, the remaining burn-in and all samples are one launch each."""
    from bayes_vi_amd.inference.mcmc.device_hmc import FusedNoUTurnSamplerKernel
    rng = np.random.default_rng(1)
    y = rng.normal(7, 3, 20).astype(np.float32)
    model, ds = normal_setup(bvi, y)
    cfg = bvi.NoUTurnSampler(stepsize_adaptation_kernel=bvi.DualAveragingStepSizeAdaptation(num_adaptation_steps=800))
    mcmc = bvi.MCMC(model=model, dataset=ds, transition_kernel=cfg)
    kernel, trace_fns, names = cfg(mcmc.posterior.unconstrained_log_prob_and_grad)
    assert isinstance(kernel.inner, FusedNoUTurnSamplerKernel)
    res = mcmc.fit(num_chains=2, num_samples=4000, num_burnin_steps=1000, progress_bar=False, seed=5)
    loc, scale = (s.cpu().numpy() for s in res.samples)
    assert loc.shape == (4000, 2) and set(res.trace.keys()) == set(names)
    ml, sl, ms, ss = exact_normal_moments(y.astype(np.float64))
    assert loc.mean() == pytest.approx(ml, abs=5 * sl / math.sqrt(8000 / 1.5) + 0.02 * sl)
    assert scale.mean() == pytest.approx(ms, abs=5 * ss / math.sqrt(8000 / 1.5) + 0.02 * ss)
    assert loc.std() == pytest.approx(sl, rel=0.08) and scale.std() == pytest.approx(ss, rel=0.10)
    assert not bool(res.trace['has_divergence'].any())
    assert 1 <= int(res.trace['leapfrogs_taken'].min()) and int(res.trace['leapfrogs_taken'].max()) <= 31
    assert 0.6 < float(torch.exp(res.trace['log_accept_ratio']).mean()) < 0.95


def test_fullcov_mixture_posterior_and_nuts(bvi, gpu_device):
    """BASELINE.json configs[3] in miniature: full-covariance mixture, FillScaleTriL-constrained Cholesky
    factors, posterior assembly against the oracle and a short NUTS run."""
    import bayes_vi_amd.bijectors as tfb
    rng = np.random.default_rng(6)
    k, d, n, c = 3, 4, 600, 4
    m = d * (d + 1) // 2
    true_mu = np.array([[-6.0] * d, [0.0] * d, [6.0] * d])
    comp = rng.integers(0, k, n)
    Y = (true_mu[comp] + rng.normal(size=(n, d))).astype(np.float32)
    priors = collections.OrderedDict(weights=bvi.tfd.Dirichlet(torch.ones(k)),
                                     locs=bvi.tfd.Normal(torch.zeros(k, d), 10.),
                                     scale_tril=bvi.tfd.Normal(torch.zeros(k, d, d), 5.))
    model = bvi.Model(priors, lambda weights, locs, scale_tril: bvi.tfd.MixtureSameFamily(
        mixture_distribution=bvi.tfd.Categorical(probs=weights),
        components_distribution=bvi.tfd.MultivariateNormalTriL(loc=locs, scale_tril=scale_tril)),
        constraining_bijectors=[tfb.SoftmaxCentered(), tfb.Identity(), tfb.FillScaleTriL(d, 1e-3, batch=k)])
    df = pd.DataFrame(Y, columns=['y%d' % i for i in range(d)])
    ds = bvi.make_dataset_from_df(df, target_names=list(df.columns))
    cfg = bvi.NoUTurnSampler(step_size=0.02, max_tree_depth=4)
    mc = bvi.MCMC(model, ds, cfg)
    post = mc.posterior
    # the Cholesky-factor bijector is one of the device-side blocks, so assembly and NUTS stay on the stream
    from bayes_vi_amd.inference.mcmc.device_hmc import BatchedDeviceNoUTurnSamplerKernel
    assert post.device_blocks() is not None
    made = cfg(post.unconstrained_log_prob_and_grad)[0]
    assert isinstance(getattr(made, 'inner', made), BatchedDeviceNoUTurnSamplerKernel)
    pu = model.flat_unconstrained_param_event_ndims
    assert pu == (k - 1) + k * d + k * m
    z = np.concatenate([rng.normal(0, .2, (c, k - 1)), true_mu.reshape(1, -1) + rng.normal(0, .2, (c, k * d)),
                        rng.normal(0, .2, (c, k * m)) + 0.3], -1).astype(np.float32)
    lp, g = post.unconstrained_log_prob_and_grad(torch.tensor(z, device=gpu_device))
    # oracle: same model with one FillScaleTriL block per component; priors on every entry of the
    # [K, D, D] parameter (the structurally zero upper entries add the same constant on both sides)
    blocks = [dict(name='w', size=k, prior=('dirichlet', np.ones(k)), bijector=cf.SoftmaxCentered()),
              dict(name='mu', size=k * d, prior=('normal', 0.0, 10.0), bijector=cf.Identity())]
    blocks += [dict(name='L%d' % i, size=d * d, prior=('normal', 0.0, 5.0), bijector=cf.FillScaleTriL(d, 1e-3))
               for i in range(k)]

    def lik(th):
        w, mu, L = th[:, :k], th[:, k:k + k * d].reshape(-1, k, d), th[:, k + k * d:].reshape(-1, k, d, d)
        lpv, (gw, gmu, gL) = cf.gmm_fullcov(Y, w, mu, L)
        return lpv, np.concatenate([gw, gmu.reshape(len(th), -1), gL.reshape(len(th), -1)], -1)
    spec = cf.ModelSpec(blocks, lik)
    rlp, rg = spec.log_post_unconstrained(z)
    assert_logp_close(lp.cpu().numpy(), rlp)
    assert_grad_close(g.cpu().numpy(), rg, term_scale=np.full_like(rg, 3.0 * n))
    init = [a.clone() for a in model.split_state(post.bijector.forward(torch.tensor(z)))]
    res = mc.fit(initial_state=init, num_samples=30, num_burnin_steps=20, progress_bar=False, seed=3)
    assert [tuple(s.shape) for s in res.samples] == [(30, c, k), (30, c, k, d), (30, c, k, d, d)]
    assert all(bool(torch.isfinite(s).all()) for s in res.samples)
    assert float(res.accept_ratios.mean()) > 0.3


@pytest.mark.parametrize('k,d', [(2, 12), (4, 24)])
def test_fullcov_device_assembly_and_hmc_step_wide_parameter_vectors(bvi, gpu_device, k, d):
    """Parameter vectors of a few hundred / a few thousand elements (256 / 1024 threads per chain in the
    state kernels): posterior assembly with the FillScaleTriL block, and one HMC transition through
    bvi_hmc_begin / bvi_leapfrog_pre / _post / bvi_hmc_end with injected momenta, against the oracle."""
    import bayes_vi_amd.bijectors as tfb
    from bayes_vi_amd.inference.mcmc.device_hmc import DeviceHamiltonianMonteCarloKernel
    from oracle import leapfrog as lf
    rng = np.random.default_rng(60 + d)
    n, c = 300, 3
    m = d * (d + 1) // 2
    true_mu = rng.integers(-8, 9, (k, d)).astype(np.float64)
    comp = rng.integers(0, k, n)
    Y = (true_mu[comp] + rng.normal(size=(n, d))).astype(np.float32)
    priors = collections.OrderedDict(weights=bvi.tfd.Dirichlet(torch.ones(k) * 2.0),
                                     locs=bvi.tfd.Normal(torch.zeros(k, d), 10.),
                                     scale_tril=bvi.tfd.Normal(torch.zeros(k, d, d), 5.))
    model = bvi.Model(priors, lambda weights, locs, scale_tril: bvi.tfd.MixtureSameFamily(
        mixture_distribution=bvi.tfd.Categorical(probs=weights),
        components_distribution=bvi.tfd.MultivariateNormalTriL(loc=locs, scale_tril=scale_tril)),
        constraining_bijectors=[tfb.SoftmaxCentered(), tfb.Identity(), tfb.FillScaleTriL(d, 1e-3, batch=k)])
    df = pd.DataFrame(Y, columns=['y%d' % i for i in range(d)])
    ds = bvi.make_dataset_from_df(df, target_names=list(df.columns))
    post = bvi.MCMC(model, ds).posterior
    assert post.device_blocks() is not None
    pu = (k - 1) + k * d + k * m
    # start near identity Cholesky factors: packed entries ~ 0 off the diagonal, softplus^-1(1) on it
    zL = tfb.FillScaleTriL(d, 1e-3, batch=k).inverse(torch.eye(d).repeat(k, 1, 1).reshape(-1)).numpy()
    z = np.concatenate([rng.normal(0, .2, (c, k - 1)), true_mu.reshape(1, -1) + rng.normal(0, .1, (c, k * d)),
                        zL[None] + rng.normal(0, .03, (c, k * m))], -1).astype(np.float32)
    assert z.shape == (c, pu)
    blocks = [dict(name='w', size=k, prior=('dirichlet', np.ones(k) * 2.0), bijector=cf.SoftmaxCentered()),
              dict(name='mu', size=k * d, prior=('normal', 0.0, 10.0), bijector=cf.Identity())]
    blocks += [dict(name='L%d' % i, size=d * d, prior=('normal', 0.0, 5.0), bijector=cf.FillScaleTriL(d, 1e-3))
               for i in range(k)]

    def lik(th):
        w, mu, L = th[:, :k], th[:, k:k + k * d].reshape(-1, k, d), th[:, k + k * d:].reshape(-1, k, d, d)
        lpv, (gw, gmu, gL) = cf.gmm_fullcov(Y, w, mu, L)
        return lpv, np.concatenate([gw, gmu.reshape(len(th), -1), gL.reshape(len(th), -1)], -1)
    spec = cf.ModelSpec(blocks, lik)
    zt = torch.tensor(z, device=gpu_device)
    lp, g = post.unconstrained_log_prob_and_grad(zt)
    rlp, rg = spec.log_post_unconstrained(z)
    assert_logp_close(lp.cpu().numpy(), rlp)
    assert_grad_close(g.cpu().numpy(), rg, term_scale=np.full_like(rg, 3.0 * n))
    # the generic composition from the bijector objects gives the same numbers
    saved, post._blocks = post._blocks, None
    lp_t, g_t = post.unconstrained_log_prob_and_grad(zt)
    post._blocks = saved
    assert_logp_close(lp_t.cpu().numpy(), rlp)
    np.testing.assert_allclose(g.cpu().numpy(), g_t.cpu().numpy(), rtol=2e-4, atol=2e-3 * np.abs(rg).max())
    eps, L = 2e-3, 3
    kern = DeviceHamiltonianMonteCarloKernel(post, step_size=eps, num_leapfrog_steps=L, seed=7)
    p = rng.normal(size=(c, pu)).astype(np.float32); lu = np.log(rng.uniform(size=c)).astype(np.float32)
    z1, st1, tr = kern.one_step(zt, kern.bootstrap(zt), momenta=torch.tensor(p, device=gpu_device),
                                log_uniform=torch.tensor(lu, device=gpu_device))
    q_ref, acc_ref, lr_ref, lp_ref = lf.hmc_step(spec.log_post_unconstrained, z, p, lu, eps, L)
    safe = np.abs(lu - lr_ref) > 2e-2
    assert safe.any()
    np.testing.assert_array_equal(tr['is_accepted'].cpu().numpy()[safe], acc_ref[safe])
    np.testing.assert_allclose(z1.cpu().numpy()[safe], q_ref[safe], rtol=5e-5, atol=5e-5)
    np.testing.assert_allclose(tr['log_accept_ratio'].cpu().numpy(), lr_ref, atol=2e-2 + 1e-6 * np.abs(rlp).max())


# ---- LeapfrogIntegrator.solve and HamiltonianFlow on the device (symplectic_integrators.py:12-59,
# ---- flow_bijectors.py:67-81)
def test_leapfrog_solve_on_device_orbit_and_step_size_forms(gpu_device):
    from bayes_vi_amd.utils.symplectic_integrators import LeapfrogIntegrator
    from oracle import leapfrog as lf
    integ = LeapfrogIntegrator()
    U = lambda q: 0.5 * (q * q).sum(-1)
    T = lambda p: 0.5 * (p * p).sum(-1)
    # README.md:571-591: harmonic oscillator, step 0.5, from (1, 0): 20 applications of one step
    x = torch.tensor([[1.0, 0.0]], device=gpu_device)
    orbit = []
    for _ in range(20):
        x = integ.solve(U, T, x, torch.tensor(0.5, device=gpu_device), 1)
        orbit.append(x.cpu().numpy()[0])
    np.testing.assert_allclose(orbit[0], [0.875, -0.46875], rtol=1e-7)
    np.testing.assert_allclose(orbit[-1], [-0.7760410416, 0.6106556172], rtol=2e-5)
    inv = [q * q + p * p / (1 - 0.25 * 0.25) for q, p in orbit]
    np.testing.assert_allclose(inv, 1.0, rtol=1e-5)
    # per-dimension, negative and trainable step sizes against the fp64 restatement
    rng = np.random.default_rng(0)
    d = 5
    A = rng.normal(size=(d, d)); A = A @ A.T / d + np.eye(d)
    At = torch.tensor(A, dtype=torch.float32, device=gpu_device)
    Uq = lambda q: 0.5 * ((q @ At) * q).sum(-1) + 0.1 * (q ** 4).sum(-1)
    gU = lambda q: q @ A + 0.4 * q ** 3
    x0 = rng.normal(size=(7, 2 * d)).astype(np.float32)
    for eps in (np.float32(0.07), -np.linspace(0.02, 0.1, d).astype(np.float32)):
        e = torch.nn.Parameter(torch.tensor(eps, device=gpu_device))
        out = integ.solve(Uq, T, torch.tensor(x0, device=gpu_device), e, 4)
        ref = lf.solve(gU, lambda p: p, x0, eps, 4)
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=2e-5, atol=2e-5)
        out.sum().backward()
        assert e.grad is not None and bool(torch.isfinite(e.grad).all())
    with pytest.raises(ValueError):
        integ.solve(Uq, T, torch.zeros(3, 5, device=gpu_device), 0.1, 1)


def test_hamiltonian_flow_is_volume_preserving_and_invertible_on_device(bvi, gpu_device):
    torch.manual_seed(0)
    d = 6
    flow = bvi.HamiltonianFlow(d, step_sizes=torch.nn.Parameter(torch.tensor(0.1)), num_integration_steps=3,
                               hidden_layers=[32, 32]).to(gpu_device)
    x = torch.randn(16, 2 * d, device=gpu_device)
    y = flow.forward(x)
    assert tuple(y.shape) == (16, 2 * d) and float((y - x).detach().abs().max()) > 1e-3
    back = flow.inverse(y)
    np.testing.assert_allclose(back.detach().cpu().numpy(), x.cpu().numpy(), atol=1e-5, rtol=1e-5)
    assert float(flow.forward_log_det_jacobian(x).abs().max()) == 0.0
    assert float(flow.inverse_log_det_jacobian(y).abs().max()) == 0.0
    # the declared log-det-Jacobian of 0 is the true one: the Jacobian of the flow map has determinant 1
    J = torch.autograd.functional.jacobian(lambda v: flow.forward(v[None])[0], x[0].double().float())
    assert float(torch.linalg.det(J.double())) == pytest.approx(1.0, abs=2e-4)


def test_leapfrog_solve_fast_path_on_the_model_posterior(bvi, gpu_device):
    """`solve` with U = -log posterior and T = |p|^2/2 runs on the device leapfrog kernels around the fused pass;
    same result as the generic route (closed-form gradient hook, torch updates) and as the fp64 restatement."""
    from bayes_vi_amd.fused_target import PotentialEnergy, QuadraticKineticEnergy
    from bayes_vi_amd.utils.symplectic_integrators import LeapfrogIntegrator
    from oracle import leapfrog as lf
    rng = np.random.default_rng(5)
    n, d, c = 4000, 12, 9
    X = rng.normal(size=(n, d)).astype(np.float32); w = rng.normal(size=d)
    y = (0.5 + X @ w + rng.normal(size=n)).astype(np.float32)
    priors = collections.OrderedDict(b=bvi.tfd.Normal(0., 10.), w=bvi.tfd.Normal(torch.zeros(d), 10.),
                                     scale=bvi.tfd.HalfNormal(10.))
    model = bvi.Model(priors, lambda b, w, scale, features: bvi.tfd.Normal(
        loc=b + bvi.symbolic.matvec(features, w), scale=scale))
    df = pd.DataFrame(X, columns=['f%d' % i for i in range(d)]); df['y'] = y
    ds = bvi.make_dataset_from_df(df, ['y'], ['f%d' % i for i in range(d)], format_features_as='tensor')
    post = bvi.MCMC(model, ds).posterior
    spec = cf.linreg_model(X, y)
    pu = d + 2
    z0 = np.concatenate([0.5 + 0.05 * rng.normal(size=(c, 1)), w + 0.05 * rng.normal(size=(c, d)),
                         0.5 + 0.05 * rng.normal(size=(c, 1))], -1)
    x0 = np.concatenate([z0, rng.normal(size=(c, pu))], -1).astype(np.float32)
    U, T = PotentialEnergy(post), QuadraticKineticEnergy()
    integ = LeapfrogIntegrator()
    xt = torch.tensor(x0, device=gpu_device)
    eps, steps = 2e-3, 5
    assert integ._device_route(U, T, xt, eps) is not None
    fast = integ.solve(U, T, xt, eps, steps)
    assert integ._device_route(U, T, xt, torch.full((pu,), eps, device=gpu_device)) is None
    generic = integ.solve(U, T, xt, torch.full((pu,), eps, device=gpu_device), steps)
    ref = lf.solve(lambda q: -spec.log_post_unconstrained(q)[1], lambda p: p, x0, eps, steps)
    scale = np.abs(ref).max()
    np.testing.assert_allclose(fast.cpu().numpy(), ref, rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(generic.cpu().numpy(), ref, rtol=0, atol=2e-5 * scale)
    assert torch.equal(xt, torch.tensor(x0, device=gpu_device))          # the input state is not modified
    # the energy itself, and its autograd gradient through the fused node
    q = xt[:, :pu].clone().requires_grad_(True)
    e = U(q)
    rlp, rg = spec.log_post_unconstrained(x0[:, :pu])
    assert_logp_close(-e.detach().cpu().numpy(), rlp)
    e.sum().backward()
    assert_grad_close(-q.grad.cpu().numpy(), rg)
