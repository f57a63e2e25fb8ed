"""CPU: pins of oracle/elbo.py (the fp64 restatement of one VI.fit optimisation step).

The reference ships no fixture for this path and TFP cannot run here, so the restatement is pinned
internally: closed-form ADVI gradients against fp64 autograd of the same loss and against central
differences, the Adam update against a hand-computed sequence, the leapfrog of the flow against the
exact harmonic-oscillator orbit of the reference's README.md:571-591, and volume preservation of the
Hamiltonian flow (log|det J| = 0, flow_bijectors.py:77-81) against an autograd Jacobian.
"""
import math

import numpy as np
import pytest
import torch

from oracle import closed_form as cf
from oracle import elbo


def _linreg(n=200, d=3, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, d)); w = rng.normal(size=d)
    y = 0.5 + X @ w + rng.normal(size=n)
    return X, y, cf.linreg_model(X, y)


@pytest.mark.parametrize('mean_field', [True, False])
def test_advi_closed_form_gradients_match_autograd_and_differences(mean_field):
    X, y, spec = _linreg()
    rng = np.random.default_rng(1)
    pu = spec.pu
    params = {'loc': rng.normal(size=pu)}
    if mean_field:
        params['raw_scale'] = elbo.softplus_inverse(np.ones(pu)) + 0.3 * rng.normal(size=pu)
    else:
        raw = np.tril(0.2 * rng.normal(size=(pu, pu)), -1)
        raw[np.diag_indices(pu)] = elbo.softplus_inverse(np.full(pu, 1.0 - 1e-5)) + 0.2 * rng.normal(size=pu)
        params['raw_tril'] = raw
    eps = rng.normal(size=(7, pu))
    loss, grads, aux = elbo.advi_loss_and_grads(spec, params, eps, mean_field)
    # the per-sample pieces: log q is the base density minus the constraining Jacobian, the target the closed form
    lp, _ = spec.log_post_constrained(aux['theta'])
    np.testing.assert_allclose(aux['log_target'], lp, rtol=1e-12)
    assert loss == pytest.approx((aux['logq'] - lp).mean(), rel=1e-13)
    # autograd twin
    post = elbo.t_log_posterior('linreg', (X, y), spec.blocks)
    fn = elbo.t_advi_loss(spec.blocks, post, mean_field)
    tl, tg = elbo.t_loss_and_grads(lambda v: fn(v, eps), params)
    assert tl == pytest.approx(loss, rel=1e-12)
    for k in grads:
        np.testing.assert_allclose(grads[k], tg[k], rtol=1e-9, atol=1e-9 * np.abs(tg[k]).max())
    # central differences on a few coordinates
    for k in grads:
        flat = np.asarray(params[k], np.float64).reshape(-1)
        idxs = [i for i in range(flat.size)
                if not (k == 'raw_tril' and (i // pu) < (i % pu))][:6]      # upper triangle is unused
        for i in idxs:
            h = 1e-6
            up = dict(params); dn = dict(params)
            a = flat.copy(); a[i] += h; up[k] = a.reshape(np.shape(params[k]))
            b = flat.copy(); b[i] -= h; dn[k] = b.reshape(np.shape(params[k]))
            fd = (elbo.advi_loss_and_grads(spec, up, eps, mean_field)[0]
                  - elbo.advi_loss_and_grads(spec, dn, eps, mean_field)[0]) / (2 * h)
            assert fd == pytest.approx(np.asarray(grads[k]).reshape(-1)[i], rel=2e-5, abs=1e-6)
    if not mean_field:
        assert np.all(np.triu(grads['raw_tril'], 1) == 0)


def test_initial_advi_state_has_unit_scale():
    # surrogate_posteriors.py:116-119: TransformedVariable(ones, Softplus); :126-130: eye through FillScaleTriL
    pu = 4
    assert np.allclose(elbo.softplus(elbo.softplus_inverse(np.ones(pu))), 1.0)
    raw = np.zeros((pu, pu)); raw[np.diag_indices(pu)] = elbo.softplus_inverse(np.full(pu, 1.0 - 1e-5))
    np.testing.assert_allclose(elbo.advi_scale_tril(raw), np.eye(pu), atol=1e-12)


def test_non_finite_discrepancies_are_dropped_from_loss_and_gradient():
    y = np.random.default_rng(2).normal(7, 3, 20)
    spec = cf.normal_model(y)
    params = {'loc': np.array([6.0, 1.0]), 'raw_scale': np.array([0.0, 0.0])}
    eps = np.random.default_rng(3).normal(size=(5, 2))
    eps[2, 1] = -2000.0            # scale = softplus(-1400) underflows to 0: log-density is not finite
    with np.errstate(all='ignore'):
        loss, grads, aux = elbo.advi_loss_and_grads(spec, params, eps, True)
    assert not aux['finite'][2] and aux['finite'].sum() == 4
    keep = np.array([0, 1, 3, 4])
    loss4, grads4, _ = elbo.advi_loss_and_grads(spec, params, eps[keep], True)
    assert np.isfinite(loss) and loss == pytest.approx(loss4, rel=1e-13)
    for k in grads:
        np.testing.assert_allclose(grads[k], grads4[k], rtol=1e-12)
    m, finite, n = elbo.masked_mean([1.0, np.inf, 3.0, np.nan])
    assert m == 2.0 and n == 2 and list(finite) == [True, False, True, False]


def test_clip_by_norm_and_adam_known_answers():
    g = np.array([3.0, 4.0])
    np.testing.assert_allclose(elbo.clip_by_norm(g), [0.6, 0.8])
    np.testing.assert_allclose(elbo.clip_by_norm(0.1 * g), 0.1 * g)
    np.testing.assert_array_equal(elbo.clip_by_norm(np.zeros(3)), np.zeros(3))
    # two Adam steps by hand (b1 = 0.9, b2 = 0.999, lr = 0.1, epsilon = 1e-7)
    opt = elbo.Adam({'x': np.zeros(1)}, learning_rate=0.1)
    x1 = opt.step({'x': np.array([1.0])}, {'x': np.array([0.5])})['x']
    m1, v1 = 0.05, 0.00025
    lr1 = 0.1 * math.sqrt(1 - 0.999) / (1 - 0.9)
    assert x1[0] == pytest.approx(1.0 - lr1 * m1 / (math.sqrt(v1) + 1e-7), rel=1e-14)
    x2 = opt.step({'x': x1}, {'x': np.array([-0.25])})['x']
    m2 = 0.9 * m1 + 0.1 * -0.25; v2 = 0.999 * v1 + 0.001 * 0.0625
    lr2 = 0.1 * math.sqrt(1 - 0.999 ** 2) / (1 - 0.81)
    assert x2[0] == pytest.approx(x1[0] - lr2 * m2 / (math.sqrt(v2) + 1e-7), rel=1e-14)
    # the first step moves every coordinate by ~lr against the sign of its gradient
    assert x1[0] == pytest.approx(0.9, abs=1e-4)


def test_flow_leapfrog_reproduces_readme_orbit_and_preserves_volume():
    # README.md:571-591: U = q^2/2, T = p^2/2, step 0.5, one step per application, from (1, 0)
    U = lambda q: 0.5 * (q * q).sum(-1)
    T = lambda p: 0.5 * (p * p).sum(-1)
    x = torch.tensor([[1.0, 0.0]], dtype=torch.float64)
    x1 = elbo.t_leapfrog(U, T, x, 0.5, 1)
    np.testing.assert_allclose(x1.detach().numpy(), [[0.875, -0.46875]], rtol=1e-15)
    for _ in range(19):
        x1 = elbo.t_leapfrog(U, T, x1, 0.5, 1)
    np.testing.assert_allclose(x1.detach().numpy(), [[-0.7760410416, 0.6106556172]], rtol=1e-9)
    # MLP energies: the flow map has unit Jacobian determinant and negating the step inverts it
    rng = np.random.default_rng(0)
    d, h = 3, 5
    layers = lambda: [(torch.tensor(rng.normal(size=(d, h)) * 0.7), torch.tensor(rng.normal(size=h) * 0.1)),
                      (torch.tensor(rng.normal(size=(h, 1)) * 0.7), torch.tensor(rng.normal(size=1) * 0.1))]
    lu, lt = layers(), layers()
    Um = lambda q: elbo.t_mlp(q, lu, 'tanh', 'softplus')
    Tm = lambda p: elbo.t_mlp(p, lt, 'tanh', 'softplus')
    x0 = torch.tensor(rng.normal(size=(1, 2 * d)))
    f = lambda v: elbo.t_leapfrog(Um, Tm, v[None, :], 0.3, 2)[0]
    J = torch.autograd.functional.jacobian(f, x0[0])
    assert float(torch.linalg.det(J)) == pytest.approx(1.0, abs=1e-10)
    back = elbo.t_leapfrog(Um, Tm, elbo.t_leapfrog(Um, Tm, x0, 0.3, 2), -0.3, 2)
    np.testing.assert_allclose(back.detach().numpy(), x0.numpy(), atol=1e-12)


def test_flow_surrogate_with_identity_flow_reduces_to_a_standard_normal_surrogate():
    """With zero step sizes the Hamiltonian flow is the identity, so the augmented-flow loss must equal
    E[log N(x) - fldj - log pi(theta) - log N(a)] computed by hand (surrogate_posteriors.py:79-88, 190-205)."""
    X, y, spec = _linreg(n=50, d=2, seed=4)
    pu = spec.pu
    rng = np.random.default_rng(5)
    fs = elbo.FlowSurrogate(spec.blocks, n_flows=2, n_steps=2, n_energy_layers=2)
    v = {}
    for i in range(2):
        v['flow%d.step' % i] = np.array(0.0)
        for e in 'UT':
            v['flow%d.%s.0.k' % (i, e)] = rng.normal(size=(pu, 4)); v['flow%d.%s.0.b' % (i, e)] = rng.normal(size=4)
            v['flow%d.%s.1.k' % (i, e)] = rng.normal(size=(4, 1)); v['flow%d.%s.1.b' % (i, e)] = rng.normal(size=1)
    noise = rng.normal(size=(6, 2 * pu))
    post = elbo.t_log_posterior('linreg', (X, y), spec.blocks)
    loss, _ = elbo.t_loss_and_grads(lambda leaves: fs.loss(leaves, post, noise), v)
    z, a = noise[:, :pu], noise[:, pu:]
    lp_u, _ = spec.log_post_unconstrained(z)
    logn = lambda t: (-0.5 * t * t - 0.5 * elbo.LN2PI).sum(-1)
    by_hand = (logn(noise) - lp_u - logn(a)).mean()
    assert loss == pytest.approx(by_hand, rel=1e-12)
