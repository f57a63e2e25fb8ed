"""Pins the CPU oracle before anything is compared against it.

The reference ships no tests, so the pins are: scipy.stats / torch.distributions
cross-checks, published known answers, fp64 central differences, and the exact
harmonic-oscillator leapfrog orbit of the reference's README.md:571-591.
"""
import math

import numpy as np
import pytest
import scipy.stats as st
import torch

from oracle import closed_form as cf
from oracle import leapfrog as lf
from oracle import restated_torch as rt


def _fd(f, x, eps=1e-6):
    g = np.zeros_like(x)
    for idx in np.ndindex(*x.shape):
        xp = x.copy(); xm = x.copy()
        xp[idx] += eps; xm[idx] -= eps
        g[idx] = (f(xp)[idx[0]] - f(xm)[idx[0]]) / (2 * eps)
    return g


def test_known_answers():
    y = np.array([6.1, 7.9, 3.2, 10.4])
    lp, _ = cf.normal_iid(y, np.array([[6.8, 2.4]]))
    assert lp[0] == pytest.approx(-9.575198526678735, abs=1e-12)
    lp, _ = cf.halfnormal_prior(2.4, 10.0)
    assert float(lp) == pytest.approx(-2.557176445638773, abs=1e-12)


def test_priors_vs_scipy_and_torch():
    x = np.array([-1.3, 0.2, 4.0])
    lp, g = cf.normal_prior(x, 0.5, 10.0)
    np.testing.assert_allclose(lp, st.norm(0.5, 10.0).logpdf(x), rtol=1e-13)
    xs = np.array([0.3, 2.4, 11.0])
    lp, _ = cf.halfnormal_prior(xs, 10.0)
    np.testing.assert_allclose(lp, st.halfnorm(scale=10.0).logpdf(xs), rtol=1e-13)
    np.testing.assert_allclose(
        lp, torch.distributions.HalfNormal(torch.tensor(10.0, dtype=torch.float64)).log_prob(torch.tensor(xs)).numpy(),
        rtol=1e-13)
    w = np.array([0.2, 0.5, 0.3]); a = np.array([1.5, 2.0, 0.7])
    lp, _ = cf.dirichlet_prior(w, a)
    assert float(lp) == pytest.approx(st.dirichlet(a).logpdf(w), rel=1e-12)
    lp, _ = cf.dirichlet_prior(w, np.ones(3))
    assert float(lp) == pytest.approx(math.lgamma(3), rel=1e-12)


def test_normal_iid_vs_scipy_and_fd():
    rng = np.random.default_rng(0)
    y = rng.normal(7, 3, 50)
    th = np.array([[6.5, 2.8], [0.1, 9.0], [7.2, 0.9]])
    lp, g = cf.normal_iid(y, th)
    ref = np.array([st.norm(t[0], t[1]).logpdf(y).sum() for t in th])
    np.testing.assert_allclose(lp, ref, rtol=1e-12)
    np.testing.assert_allclose(g, _fd(lambda t: cf.normal_iid(y, t)[0], th), rtol=1e-6, atol=1e-6)


def test_linreg_vs_scipy_and_fd():
    rng = np.random.default_rng(1)
    X = rng.normal(size=(40, 5)); y = rng.normal(size=40)
    th = np.concatenate([rng.normal(size=(3, 6)), rng.uniform(0.5, 2, (3, 1))], -1)
    lp, g = cf.linreg(X, y, th)
    ref = np.array([st.norm(t[0] + X @ t[1:6], t[6]).logpdf(y).sum() for t in th])
    np.testing.assert_allclose(lp, ref, rtol=1e-12)
    np.testing.assert_allclose(g, _fd(lambda t: cf.linreg(X, y, t)[0], th), rtol=1e-6, atol=1e-6)


def test_gmm1d_vs_torch_and_fd():
    rng = np.random.default_rng(2)
    k = 3
    y = np.concatenate([rng.normal(-5, 1, 30), rng.normal(4, 2, 30)])
    th = np.concatenate([rng.dirichlet(np.ones(k), 2), rng.normal(0, 4, (2, k)), rng.uniform(0.5, 3, (2, k))], -1)
    lp, g = cf.gmm1d(y, th, k)
    for c in range(2):
        t = torch.tensor(th[c])
        mix = torch.distributions.MixtureSameFamily(
            torch.distributions.Categorical(probs=t[:k]),
            torch.distributions.Normal(t[k:2 * k], t[2 * k:]))
        assert lp[c] == pytest.approx(mix.log_prob(torch.tensor(y)).sum().item(), rel=1e-12)
    np.testing.assert_allclose(g, _fd(lambda t: cf.gmm1d(y, t, k)[0], th), rtol=2e-6, atol=2e-6)
    # un-normalised weights: Categorical(probs=w) renormalises, so scaling w leaves logp unchanged
    th2 = th.copy(); th2[:, :k] *= 3.0
    np.testing.assert_allclose(cf.gmm1d(y, th2, k)[0], lp, rtol=1e-12)


def test_gmm_fullcov_vs_torch_and_fd():
    rng = np.random.default_rng(3)
    c, k, d, n = 2, 3, 4, 25
    Y = rng.normal(size=(n, d)) * 2
    w = rng.dirichlet(np.ones(k), c); mu = rng.normal(size=(c, k, d))
    L = np.tril(rng.normal(size=(c, k, d, d)) * 0.3)
    for i in range(d):
        L[..., i, i] = rng.uniform(0.7, 1.5, (c, k))
    lp, (gw, gmu, gL) = cf.gmm_fullcov(Y, w, mu, L)
    for ci in range(c):
        mix = torch.distributions.MixtureSameFamily(
            torch.distributions.Categorical(probs=torch.tensor(w[ci])),
            torch.distributions.MultivariateNormal(torch.tensor(mu[ci]), scale_tril=torch.tensor(L[ci])))
        assert lp[ci] == pytest.approx(mix.log_prob(torch.tensor(Y)).sum().item(), rel=1e-11)
    wt = torch.tensor(w, requires_grad=True); mt = torch.tensor(mu, requires_grad=True)
    Lt = torch.tensor(L, requires_grad=True)
    tot = 0
    for ci in range(c):
        mix = torch.distributions.MixtureSameFamily(
            torch.distributions.Categorical(probs=wt[ci]),
            torch.distributions.MultivariateNormal(mt[ci], scale_tril=Lt[ci]))
        tot = tot + mix.log_prob(torch.tensor(Y)).sum()
    tot.backward()
    np.testing.assert_allclose(gw, wt.grad.numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(gmu, mt.grad.numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(gL, np.tril(Lt.grad.numpy()), rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize('bij,n', [(cf.Softplus(), 3), (cf.Exp(), 2), (cf.SoftmaxCentered(), 4),
                                   (cf.FillScaleTriL(3, 1e-3), 6), (cf.Identity(), 2)])
def test_bijectors_jacobians(bij, n):
    rng = np.random.default_rng(4)
    x = rng.normal(size=(2, n))
    y = bij.forward(x)
    if not isinstance(bij, cf.FillScaleTriL):
        np.testing.assert_allclose(bij.inverse(y), x, rtol=1e-9, atol=1e-9)
    # numerical Jacobian
    eps = 1e-6
    for b in range(2):
        J = np.zeros((y.shape[-1], n))
        for j in range(n):
            xp = x[b].copy(); xm = x[b].copy(); xp[j] += eps; xm[j] -= eps
            J[:, j] = (bij.forward(xp[None])[0] - bij.forward(xm[None])[0]) / (2 * eps)
        if isinstance(bij, cf.FillScaleTriL):
            rows = [i * bij.d + j for i in range(bij.d) for j in range(i + 1)]
            Jsq = J[rows]
            ref = np.linalg.slogdet(Jsq)[1]
        else:
            ref = 0.5 * np.linalg.slogdet(J.T @ J)[1]
        assert bij.fldj(x)[b] == pytest.approx(ref, abs=1e-6)
        g = rng.normal(size=y.shape[-1])
        np.testing.assert_allclose(bij.vjp(x[b:b + 1], g[None])[0], J.T @ g, rtol=1e-6, atol=1e-7)
    gf = _fd(lambda t: bij.fldj(t), x)
    np.testing.assert_allclose(bij.grad_fldj(x), gf, rtol=1e-6, atol=1e-7)


def test_fill_triangular_example():
    bij = cf.FillScaleTriL(3, 0.0)
    idx = bij.index
    x = np.arange(1, 7, dtype=float)
    m = np.where(idx >= 0, x[np.maximum(idx, 0)], 0.0)
    np.testing.assert_array_equal(m, [[4, 0, 0], [6, 5, 0], [3, 2, 1]])


@pytest.mark.parametrize('maker', ['normal', 'linreg', 'gmm'])
def test_unconstrained_posterior_gradient(maker):
    rng = np.random.default_rng(5)
    if maker == 'normal':
        m = cf.normal_model(rng.normal(7, 3, 20))
    elif maker == 'linreg':
        X = rng.normal(size=(30, 3)); m = cf.linreg_model(X, X @ [1.0, -2.0, 0.5] + 0.3 + rng.normal(size=30))
    else:
        m = cf.gmm1d_model(np.concatenate([rng.normal(-4, 1, 20), rng.normal(3, 1, 20)]), 3)
    z = rng.uniform(-1, 1, (3, m.pu))
    lp, g = m.log_post_unconstrained(z)
    np.testing.assert_allclose(g, _fd(lambda t: m.log_post_unconstrained(t)[0], z), rtol=5e-6, atol=5e-6)
    th = m.constrain(z)
    np.testing.assert_allclose(m.unconstrain(th), z, rtol=1e-8, atol=1e-8)


def test_leapfrog_harmonic_oscillator_orbit():
    """README.md:571-591: U = q^2/2, T = p^2/2, eps = 0.5, n = 1, start (1, 0)."""
    gU = lambda q: q
    gT = lambda p: p
    x = np.array([1.0, 0.0])
    path = [x]
    for _ in range(20):
        path.append(lf.solve(gU, gT, path[-1], 0.5, 1))
    path = np.stack(path)
    np.testing.assert_allclose(path[1], [0.875, -0.46875], rtol=0, atol=0)
    np.testing.assert_allclose(path[2], [0.53125, -0.8203125], rtol=0, atol=0)
    np.testing.assert_allclose(path[20], [-0.7760410416, 0.6106556172], atol=1e-9)
    inv = path[:, 0] ** 2 + path[:, 1] ** 2 / (1 - 0.5 ** 2 / 4)
    np.testing.assert_allclose(inv, 1.0, atol=1e-12)
    # inverse = same map with negated step (flow_bijectors.py:72-75)
    np.testing.assert_allclose(lf.solve(gU, gT, path[5], -0.5, 1), path[4], atol=1e-14)
    with pytest.raises(ValueError):
        lf.solve(gU, gT, np.zeros(3), 0.1, 1)


def test_hmc_step_reversible_and_detailed():
    rng = np.random.default_rng(6)
    m = cf.normal_model(rng.normal(7, 3, 20))
    z = rng.uniform(-1, 1, (4, 2)) + np.array([6.0, 1.0])
    p = rng.normal(size=(4, 2))
    q1, p1, lp1, _ = lf.hmc_trajectory(m.log_post_unconstrained, z, p, 0.05, 7)
    q0, p0, _, _ = lf.hmc_trajectory(m.log_post_unconstrained, q1, -p1, 0.05, 7)
    np.testing.assert_allclose(q0, z, atol=1e-10)
    np.testing.assert_allclose(-p0, p, atol=1e-10)
    qn, acc, lr, lpn = lf.hmc_step(m.log_post_unconstrained, z, p, np.log(rng.uniform(size=4)), 0.002, 7)
    assert np.all(np.abs(lr) < 0.05)         # small step => small energy error
    assert qn.shape == z.shape and acc.dtype == bool


def test_restated_torch_matches_closed_form():
    rng = np.random.default_rng(7)
    y = rng.normal(7, 3, 200).astype(np.float32)
    th = np.array([[6.5, 2.8], [7.5, 3.3]], dtype=np.float32)
    lp, g = cf.normal_iid(y, th)
    v, gr = rt.normal_iid(torch.tensor(y), torch.tensor(th))
    np.testing.assert_allclose(v.numpy(), lp, rtol=1e-5)
    np.testing.assert_allclose(gr.numpy(), g, rtol=1e-4, atol=1e-4)
    X = rng.normal(size=(300, 8)).astype(np.float32)
    yy = (X @ rng.normal(size=8) + rng.normal(size=300)).astype(np.float32)
    th = np.concatenate([rng.normal(size=(3, 9)), rng.uniform(0.8, 2, (3, 1))], -1).astype(np.float32)
    lp, g = cf.linreg(X, yy, th)
    v, gr = rt.linreg(torch.tensor(X), torch.tensor(yy), torch.tensor(th), chunk=128)
    np.testing.assert_allclose(v.numpy(), lp, rtol=1e-5)
    np.testing.assert_allclose(gr.numpy(), g, rtol=1e-4, atol=1e-3)
    k = 4
    ym = np.concatenate([rng.normal(-5, 1, 100), rng.normal(4, 2, 100)]).astype(np.float32)
    th = np.concatenate([rng.dirichlet(np.ones(k), 2), rng.normal(0, 4, (2, k)), rng.uniform(0.5, 3, (2, k))], -1).astype(np.float32)
    lp, g = cf.gmm1d(ym, th, k)
    v, gr = rt.gmm1d(torch.tensor(ym), torch.tensor(th), k, chunk=64)
    np.testing.assert_allclose(v.numpy(), lp, rtol=1e-5)
    np.testing.assert_allclose(gr.numpy(), g, rtol=1e-4, atol=1e-3)


def test_sufficient_statistic_posterior_table():
    """univariate_gaussian_experiment/tables: NUTS posterior mean / sd for N = 100
    (dataset_infos.txt:6-15 vs mcmc_summary_statistics.txt:6-25, skipping the
    stuck-chain datasets 5 and 7).  The posterior depends on the data only
    through (N, sample loc, sample scale); its moments are computed here by
    quadrature of the oracle's log posterior on a grid and must reproduce the
    table to its two printed decimals (table inputs are themselves rounded)."""
    table = {1: (3.37, 1.65, 3.37, 0.16, 1.68, 0.12), 2: (-2.03, 2.45, -2.03, 0.25, 2.49, 0.18),
             4: (4.59, 3.69, 4.59, 0.38, 3.76, 0.27), 9: (13.94, 1.82, 13.94, 0.18, 1.85, 0.13)}
    for _, (m, s, pm, psd, sm, ssd) in table.items():
        rng = np.random.default_rng(8)
        y = rng.normal(size=100); y = (y - y.mean()) / y.std(ddof=1) * s + m
        model = cf.normal_model(y)
        locs = np.linspace(m - 8 * s / 10, m + 8 * s / 10, 201)
        scs = np.linspace(s * 0.6, s * 1.8, 241)
        L, S = np.meshgrid(locs, scs, indexing='ij')
        lp, _ = model.log_post_constrained(np.stack([L.ravel(), S.ravel()], -1))
        w = np.exp(lp - lp.max()).reshape(L.shape); w /= w.sum()
        mean_loc = (w * L).sum(); sd_loc = math.sqrt((w * (L - mean_loc) ** 2).sum())
        mean_sc = (w * S).sum(); sd_sc = math.sqrt((w * (S - mean_sc) ** 2).sum())
        assert mean_loc == pytest.approx(pm, abs=0.015)
        assert sd_loc == pytest.approx(psd, abs=0.015)
        assert mean_sc == pytest.approx(sm, abs=0.03)
        assert sd_sc == pytest.approx(ssd, abs=0.015)


LINREG_TABLE = {
    # dataset: (sample beta0, sample beta1, sample scale) from linear_regression_experiment/tables/data/
    # dataset_infos.txt:6-15, then NUTS posterior (mean, sd) of beta0, beta1, scale from
    # tables/mcmc_summary_statistics/mcmc_summary_statistics.txt:6-35.  Datasets 2 and 6 had a stuck chain
    # (acceptance 0.000 in acceptance_and_runtime_summary.txt:7, 11; r-hat 1.4) and are left out.
    1: ((-8.30, 1.39, 15.04), (-8.08, 1.54), (1.39, 0.16), (15.21, 1.11)),
    3: ((0.18, -4.75, 2.11), (0.18, 0.22), (-4.75, 0.02), (2.16, 0.16)),
    4: ((5.13, 14.30, 9.32), (5.08, 0.93), (14.30, 0.09), (9.50, 0.69)),
    5: ((17.61, 2.64, 16.11), (17.17, 1.64), (2.65, 0.19), (16.25, 1.17)),
    7: ((-8.21, 3.61, 8.52), (-8.12, 0.87), (3.61, 0.09), (8.66, 0.63)),
    8: ((13.34, -0.83, 11.55), (13.20, 1.16), (-0.83, 0.12), (11.74, 0.86)),
    9: ((1.34, -7.63, 6.94), (1.33, 0.70), (-7.63, 0.06), (7.09, 0.52)),
    10: ((-2.69, 5.07, 15.78), (-2.56, 1.55), (5.07, 0.15), (15.93, 1.13)),
}


def linreg_table_dataset(b0, b1, s, seed=0):
    """A dataset with the experiment's design (x ~ N(0, 10), 100 points; linear-regression-example.ipynb cell 19)
    whose least-squares fit is exactly (b0, b1) with residual standard deviation exactly s (1/n convention): the
    posterior depends on the data through these and the moments of x, which the reference did not save."""
    rng = np.random.default_rng(seed)
    x = rng.normal(size=100); x = (x - x.mean()) / x.std() * 10.0
    A = np.stack([np.ones_like(x), x], -1)
    e = rng.normal(size=100); e = e - A @ np.linalg.lstsq(A, e, rcond=None)[0]
    e = e / np.sqrt((e * e).mean()) * s
    return x, b0 + b1 * x + e


def test_linreg_posterior_table():
    """linear_regression_experiment/tables: the reference's NUTS posterior summaries of the 1-feature regression
    (README.md:747-756 model) are reproduced by quadrature of the oracle's log posterior on data standardised to
    the printed sample statistics.  What the unknown x cannot change is held to the table's printed decimals
    (mean of beta1, mean and sd of scale); what depends on x's realised mean and spread (beta0, sd of beta1) to
    the looseness that leaves: sd(beta1) = scale / (sd_x sqrt(n)) with sd_x within 10 +- 25 %."""
    for ds, ((b0, b1, s), (m0, s0), (m1, s1), (ms, ss)) in LINREG_TABLE.items():
        x, y = linreg_table_dataset(b0, b1, s)
        spec = cf.linreg_model(x[:, None], y)
        sb0, sb1 = s / 10.0, s / 100.0
        g0 = np.linspace(b0 - 6 * sb0, b0 + 6 * sb0, 49); g1 = np.linspace(b1 - 6 * sb1, b1 + 6 * sb1, 49)
        gs = np.linspace(s * 0.65, s * 1.7, 97)
        B0, B1, S = np.meshgrid(g0, g1, gs, indexing='ij')
        th = np.stack([B0.ravel(), B1.ravel(), S.ravel()], -1)
        lp = np.concatenate([spec.log_post_constrained(th[i:i + 40000])[0] for i in range(0, len(th), 40000)])
        w = np.exp(lp - lp.max()); w /= w.sum()
        mom = []
        for k in range(3):
            m = (w * th[:, k]).sum()
            mom.append((m, math.sqrt((w * (th[:, k] - m) ** 2).sum())))
        assert mom[1][0] == pytest.approx(m1, abs=0.011), ds
        assert mom[2][0] == pytest.approx(ms, abs=0.035), ds          # table inputs are themselves rounded
        assert mom[2][1] == pytest.approx(ss, abs=0.035), ds
        assert mom[0][0] == pytest.approx(m0, abs=0.1), ds            # shrinkage by the N(0, 10) prior included
        assert mom[0][1] == pytest.approx(s0, rel=0.06), ds
        assert mom[1][1] == pytest.approx(s1, rel=0.3, abs=0.006), ds
        # sd(beta1) / sd(beta0) = 1 / sd_x for a centred design
        assert mom[1][1] / mom[0][1] == pytest.approx(0.1, rel=0.05), ds


def test_philox_known_answers():
    """Published Philox4x32-10 known-answer vectors (Random123 kat_vectors)."""
    from oracle import philox
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox.philox4x32_10(np.array(ctr, np.uint64), np.array(key, np.uint64))
        assert tuple(int(v) for v in got) == want
    z = philox.normals(7, 3, 2000, 8)
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1) < 0.03
    assert not np.array_equal(philox.normals(7, 3, 4, 8), philox.normals(7, 4, 4, 8))
    np.testing.assert_array_equal(philox.normals(7, 3, 4, 8), philox.normals(7, 3, 4, 8))


def test_oracle_nuts_reproduces_exact_posterior_moments():
    """The oracle's No-U-Turn transition (oracle/nuts.py), iterated on the 20-point Gaussian model of
    README.md:114-162, must reproduce the exact posterior moments (quadrature of the same log posterior):
    this is what pins the restated algorithm in the absence of TFP outputs."""
    import math
    from oracle import nuts
    rng = np.random.default_rng(1)
    y = rng.normal(7, 3, 20)
    spec = cf.normal_model(y)
    f = lambda q: spec.log_post_unconstrained(q)
    z = np.array([7.0, 1.0])
    lp, g = f(z[None]); lp, g = float(lp[0]), g[0]
    draws, leaps, kin = [], [], []
    for t in range(2500):
        out = nuts.nuts_step(f, z, lp, g, chain=0, seed=11, step=t, step_size=0.35, max_tree_depth=5)
        z, lp, g = out['z'], out['lp'], out['g']
        assert out['energy'] <= lp + 1e-12            # log density minus a kinetic energy
        if t >= 300:
            draws.append(z.copy()); leaps.append(out['leapfrogs']); kin.append(lp - out['energy'])
    draws = np.array(draws)
    # `energy` is the log density minus the kinetic energy at the selected point; at stationarity the momentum
    # there is N(0, I), so the kinetic part averages P/2 = 1
    assert abs(np.mean(kin) - 1.0) < 0.1, np.mean(kin)
    loc, scale = draws[:, 0], np.log1p(np.exp(draws[:, 1]))
    m, s = y.mean(), y.std(ddof=1)
    L, S = np.meshgrid(np.linspace(m - 6 * s / math.sqrt(20), m + 6 * s / math.sqrt(20), 241),
                       np.linspace(s * 0.35, s * 3.0, 401), indexing='ij')
    lpc, _ = spec.log_post_constrained(np.stack([L.ravel(), S.ravel()], -1))
    w = np.exp(lpc - lpc.max()).reshape(L.shape); w /= w.sum()
    ml, ms = (w * L).sum(), (w * S).sum()
    sl, ss = math.sqrt((w * (L - ml) ** 2).sum()), math.sqrt((w * (S - ms) ** 2).sum())
    n_eff = len(draws) / 2.0
    assert abs(loc.mean() - ml) < 5 * sl / math.sqrt(n_eff) and abs(scale.mean() - ms) < 5 * ss / math.sqrt(n_eff)
    assert abs(loc.std() / sl - 1) < 0.1 and abs(scale.std() / ss - 1) < 0.12
    assert 1 <= min(leaps) and max(leaps) <= 31 and np.mean(leaps) > 2


# ---------------------------------------------------------------------------------------------------------------
# the reference's own mean-field ADVI results: what pins oracle/elbo.py
# ---------------------------------------------------------------------------------------------------------------
# Per dataset: the sample statistics (tables/data/dataset_infos.txt:6-15), the fitted surrogate's (mean, sd) per
# parameter over 5000 draws (tables/vi_summary_statistics/vi_summary_statistics.txt, rows "meanfield advi"), and the
# average of the last 100 losses of the 10 000-step fit that the progress bar of the experiment notebook left in its
# cell output ("avg loss", vi.py:124; univariate-gaussian-example.ipynb / linear-regression-example.ipynb cell 22).
# Settings: Adam(5e-3), 10 000 steps, sample_size 50 (notebook cells 10-11).  All ten datasets are usable here: the
# stuck NUTS chains of some datasets do not touch the VI columns used.
ADVI_GAUSS_TABLE = {
    1: ((3.37, 1.65), (3.38, 0.17), (1.67, 0.12), 200.060), 2: ((-2.03, 2.45), (-2.02, 0.25), (2.48, 0.18), 238.603),
    3: ((11.33, 0.62), (11.33, 0.06), (0.63, 0.04), 104.003), 4: ((4.59, 3.69), (4.59, 0.37), (3.75, 0.25), 278.889),
    5: ((-5.81, 7.62), (-5.74, 0.76), (7.74, 0.54), 350.186), 6: ((-17.12, 10.86), (-17.00, 1.10), (11.00, 0.75), 386.566),
    7: ((5.14, 5.47), (5.12, 0.56), (5.57, 0.40), 317.597), 8: ((-7.42, 4.24), (-7.38, 0.42), (4.30, 0.31), 292.791),
    9: ((13.94, 1.82), (13.94, 0.18), (1.84, 0.13), 210.232), 10: ((-0.84, 10.13), (-0.83, 1.02), (10.20, 0.70), 378.173),
}
ADVI_LINREG_TABLE = {
    1: ((-8.30, 1.39, 15.04), (-8.07, 1.53), (1.39, 0.16), (15.18, 1.05), 422.092),
    2: ((17.30, -4.85, 2.69), (17.30, 0.27), (-4.85, 0.03), (2.75, 0.19), 255.196),
    3: ((0.18, -4.75, 2.11), (0.19, 0.22), (-4.75, 0.02), (2.15, 0.15), 230.456),
    4: ((5.13, 14.30, 9.32), (5.08, 0.96), (14.30, 0.08), (9.48, 0.66), 375.882),
    5: ((17.61, 2.64, 16.11), (17.11, 1.60), (2.63, 0.19), (16.25, 1.11), 429.991),
    6: ((4.65, -1.31, 16.96), (4.54, 1.73), (-1.28, 0.18), (17.11, 1.18), 433.792),
    7: ((-8.21, 3.61, 8.52), (-8.14, 0.84), (3.62, 0.09), (8.70, 0.59), 366.195),
    8: ((13.34, -0.83, 11.55), (13.13, 1.13), (-0.84, 0.12), (11.73, 0.80), 396.578),
    9: ((1.34, -7.63, 6.94), (1.32, 0.69), (-7.64, 0.06), (7.09, 0.50), 346.312),
    10: ((-2.69, 5.07, 15.78), (-2.62, 1.56), (5.07, 0.15), (15.91, 1.09), 426.725),
}


def gauss_table_dataset(m, s, seed=8):
    """100 points whose mean is exactly m and whose standard deviation (1/n convention: the one under which the
    reference's printed losses are reproduced; the 1/(n-1) one misses every dataset by 0.5 = (n/2) log(n/(n-1))) is s."""
    y = np.random.default_rng(seed).normal(size=100)
    return (y - y.mean()) / y.std() * s + m


def softplus_normal_moments(mu, sd):
    """(mean, sd) of softplus(N(mu, sd)): the constrained marginal of a mean-field ADVI coordinate under Softplus."""
    from oracle import elbo
    x, w = np.polynomial.hermite_e.hermegauss(80)
    w = w / w.sum(); v = elbo.softplus(mu + sd * x); m = (w * v).sum()
    return m, math.sqrt((w * (v - m) ** 2).sum())


def oracle_advi_fit(spec, pu, seed, steps=10000, sample_size=50, lr=5e-3):
    """`VI.fit(Adam(lr), num_steps, sample_size)` with the mean-field ADVI surrogate, restated: initial loc 0 and scale 1
    (surrogate_posteriors.py:113-119 draws loc from N(0, 1); the optimum does not depend on it), oracle/elbo.py steps."""
    from oracle import elbo
    params = {'loc': np.zeros(pu), 'raw_scale': elbo.softplus_inverse(np.ones(pu))}
    opt = elbo.Adam(params, learning_rate=lr)
    rng = np.random.default_rng(seed); losses = np.empty(steps)
    tail = {k: np.zeros_like(v) for k, v in params.items()}
    for i in range(steps):
        losses[i], _, params = elbo.advi_step(spec, params, rng.normal(size=(sample_size, pu)), opt, True)
        if i >= steps - 200:                 # Adam's iterates wander around the optimum by a few learning rates:
            for k in tail:                   # the centre of the last 200 is the oracle's estimate of it
                tail[k] += params[k] / 200.0
    return tail, losses


def check_advi_row(got_mean, got_sd, want, extra_mean=0.0, extra_sd=0.0):
    """A table entry is the mean / sd of 5000 draws printed to two decimals: half a unit of the last digit, four
    standard errors of that sample (sd / sqrt(5000), sd / sqrt(10000)), four learning rates (0.02) for where the
    reference's last Adam iterate sat around the optimum (the sd moves by that fraction of itself: it is the
    softplus of such a coordinate), plus what the caller knows the rounded sample statistics can move."""
    wm, ws = want
    assert abs(got_mean - wm) <= 0.005 + 4 / math.sqrt(5000) * got_sd + 0.02 + extra_mean, (got_mean, want)
    assert abs(got_sd - ws) <= 0.005 + (4 / math.sqrt(10000) + 0.02) * got_sd + extra_sd, (got_sd, want)


def test_advi_meanfield_gaussian_table():
    """oracle/elbo.py's mean-field ADVI step, run with the experiment's settings on data standardised to the printed
    sample statistics, reproduces for all ten datasets of univariate_gaussian_experiment (a) the surrogate's mean / sd
    per parameter of tables/vi_summary_statistics/vi_summary_statistics.txt and (b) the final loss the notebook
    printed -- the absolute value of the Monte-Carlo variational loss, every normalising constant, prior and Jacobian
    term included.  (The posterior depends on the data only through n, the mean and the sd.)"""
    from oracle import elbo
    diffs = []
    for ds, ((m, s), t_loc, t_scale, t_loss) in ADVI_GAUSS_TABLE.items():
        spec = cf.normal_model(gauss_table_dataset(m, s))
        p, losses = oracle_advi_fit(spec, 2, seed=ds)
        sc = elbo.softplus(p['raw_scale'])
        # the rounded sample mean moves loc by up to 0.005, the rounded sample sd moves everything in proportion
        check_advi_row(p['loc'][0], sc[0], t_loc, extra_mean=0.005, extra_sd=0.005 / s * sc[0])
        sm, ssd = softplus_normal_moments(p['loc'][1], sc[1])
        check_advi_row(sm, ssd, t_scale, extra_mean=0.005, extra_sd=0.005 / s * ssd)
        # d loss / d s = n / s; Monte-Carlo noise of a 100-step average at S = 50 is ~0.02
        diffs.append(losses[-100:].mean() - t_loss)
        assert abs(diffs[-1]) <= 100 * 0.005 / s + 0.06, (ds, losses[-100:].mean(), t_loss)
    assert abs(np.mean(diffs)) < 0.1, diffs          # no common offset: a missing or doubled constant would show here


def test_advi_meanfield_linreg_table():
    """The same for linear_regression_experiment (model of linear-regression-example.ipynb cell 15).  The design x
    ~ N(0, 10) was not saved: what its realised mean and spread can move (intercept, sd of the slope, log-determinant
    term of the loss) gets the corresponding slack; slope mean, scale mean / sd and the loss are held tightly."""
    from oracle import elbo
    diffs = []
    for ds, ((b0, b1, s), t0, t1, ts, t_loss) in ADVI_LINREG_TABLE.items():
        x, y = linreg_table_dataset(b0, b1, s)
        p, losses = oracle_advi_fit(cf.linreg_model(x[:, None], y), 3, seed=ds)
        sc = elbo.softplus(p['raw_scale'])
        # intercept: the fitted value at x = 0 moves with the unknown mean of x (sd 1) times the slope's uncertainty
        check_advi_row(p['loc'][0], sc[0], t0, extra_mean=0.005 + 0.6 * sc[0], extra_sd=0.03 * sc[0])
        # slope: mean = least-squares slope; sd = scale / sqrt(sum x^2), sd_x within 10 +- 25 %
        check_advi_row(p['loc'][1], sc[1], t1, extra_mean=0.005, extra_sd=0.3 * sc[1])
        sm, ssd = softplus_normal_moments(p['loc'][2], sc[2])
        check_advi_row(sm, ssd, ts, extra_mean=0.005 * 1.5, extra_sd=0.005 / s * ssd)
        diffs.append(losses[-100:].mean() - t_loss)
        # n / s per unit of the rounded residual sd; log(sd_x / 10) for the unknown spread of x
        assert abs(diffs[-1]) <= 100 * 0.005 / s + 0.25 + 0.06, (ds, losses[-100:].mean(), t_loss)
    assert abs(np.mean(diffs)) < 0.15, diffs
