"""Generates tests/golden/hotpath_small.json and tests/golden/advi_step_small.json from the fp64 oracle.

The reference cannot run in this environment (TensorFlow-Probability is not
installed), so these vectors are NOT reference outputs: they are the oracle's
closed-form answers on small explicit inputs, kept so that the HIP path and any
later change of the oracle are both checked against frozen numbers.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from oracle import closed_form as cf  # noqa: E402
from oracle import elbo  # noqa: E402


def r32(a):
    return np.asarray(a, dtype=np.float32)


def main():
    rng = np.random.default_rng(20240611)
    cases = []
    # README.md:81-85 shaped data: N=20 Normal(7,3), 2 chains
    y = r32(rng.normal(7, 3, 20)); th = r32([[6.8, 2.4], [7.5, 3.5]])
    lp, g = cf.normal_iid(y, th)
    cases.append(dict(family='normal_iid', y=y.tolist(), theta=th.tolist(), logp=lp.tolist(), grad=g.tolist()))
    y = r32(rng.normal(-2, 2.4, 100)); th = r32(np.stack([rng.normal(-2, .3, 5), rng.uniform(2, 3, 5)], -1))
    lp, g = cf.normal_iid(y, th)
    cases.append(dict(family='normal_iid', y=y.tolist(), theta=th.tolist(), logp=lp.tolist(), grad=g.tolist()))
    # README.md:713-719 shaped regression: N=50, one feature
    x = r32(rng.normal(0, 10, (50, 1))); y = r32(7 + 3 * x[:, 0] + rng.normal(0, 10, 50))
    th = r32([[7.0, 3.0, 10.0], [5.0, 2.5, 8.0]])
    lp, g = cf.linreg(x, y, th)
    cases.append(dict(family='linreg', X=x.tolist(), y=y.tolist(), theta=th.tolist(), logp=lp.tolist(), grad=g.tolist()))
    for d, c, n in ((2, 3, 33), (32, 4, 70), (64, 3, 40)):
        X = r32(rng.normal(size=(n, d))); w = rng.normal(size=d)
        y = r32(0.5 + X @ w + rng.normal(size=n))
        th = r32(np.concatenate([0.5 + 0.1 * rng.normal(size=(c, 1)), w + 0.1 * rng.normal(size=(c, d)),
                                 rng.uniform(0.8, 1.5, (c, 1))], -1))
        lp, g = cf.linreg(X, y, th)
        cases.append(dict(family='linreg', X=X.tolist(), y=y.tolist(), theta=th.tolist(), logp=lp.tolist(), grad=g.tolist()))
    for k, c, n in ((3, 2, 60), (8, 4, 200)):
        locs = np.linspace(-20, 20, k); scs = 1.0 + (np.arange(k) % 3)
        comp = rng.integers(0, k, n)
        y = r32(rng.normal(locs[comp], scs[comp]))
        th = r32(np.concatenate([rng.dirichlet(np.ones(k), c), locs + rng.normal(0, .5, (c, k)),
                                 scs * rng.uniform(0.8, 1.2, (c, k))], -1))
        lp, g = cf.gmm1d(y, th, k)
        cases.append(dict(family='gmm1d', k=k, y=y.tolist(), theta=th.tolist(), logp=lp.tolist(), grad=g.tolist()))
    # full-covariance mixture (multivariate-gaussian-mixture-density-estimation-test.ipynb cell 7): D = 4, K = 3, 2 chains;
    # theta flat = (weights[K], locs[K, D], scale_tril[K, D, D]) row-major, gradient in the same layout
    d, k, c, n = 4, 3, 2, 50
    true_mu = rng.integers(-6, 7, (k, d)).astype(np.float64)
    comp = rng.integers(0, k, n)
    Y = r32(true_mu[comp] + rng.normal(size=(n, d)) * 1.5)
    w = r32(rng.dirichlet(np.ones(k) * 5, c))
    mu = r32(true_mu[None] + rng.normal(0, 0.3, (c, k, d)))
    L = np.tril(rng.normal(size=(c, k, d, d)) * 0.2)
    for i in range(d):
        L[..., i, i] = rng.uniform(1.0, 2.0, (c, k))
    L = r32(L)
    lp, (gw, gmu, gL) = cf.gmm_fullcov(Y, w, mu, L)
    theta = np.concatenate([w, mu.reshape(c, -1), L.reshape(c, -1)], -1)
    grad = np.concatenate([gw, gmu.reshape(c, -1), gL.reshape(c, -1)], -1)
    cases.append(dict(family='gmm_fullcov', k=k, d=d, Y=Y.tolist(), theta=theta.tolist(), logp=lp.tolist(),
                      grad=grad.tolist()))
    # single multivariate normal whose scale is given through a Cholesky factor of the precision
    # (multivariate-gaussian-test.ipynb cell 15): D = 3, 2 chains; theta flat = (loc[D], P[D, D]) row-major
    d, c, n = 3, 2, 40
    A = np.tril(rng.normal(size=(d, d)) * 0.4) + np.eye(d)
    Y = r32(1.0 + rng.normal(size=(n, d)) @ A.T)
    P = np.tril(rng.normal(size=(c, d, d)) * 0.15)
    for i in range(d):
        P[:, i, i] = rng.uniform(0.6, 1.4, c)
    theta = r32(np.concatenate([rng.normal(1.0, 0.3, (c, d)), P.reshape(c, -1)], -1))
    lp, grad = cf.mvn_chol_precision(Y, theta)
    cases.append(dict(family='mvn_tril', parameterisation='cholesky_solve_identity', d=d, Y=Y.tolist(),
                      theta=theta.tolist(), logp=lp.tolist(), grad=grad.tolist()))
    out = os.path.join(os.path.dirname(__file__), 'hotpath_small.json')
    with open(out, 'w') as f:
        json.dump(dict(note='closed-form fp64 oracle outputs on float32 inputs; not TFP outputs', cases=cases), f)
    print('wrote', out, os.path.getsize(out), 'bytes')
    advi_step(np.random.default_rng(20240612))


def advi_step(rng):
    """One optimisation step of VI.fit with the mean-field ADVI surrogate (vi.py:60-75, 109-116;
    surrogate_posteriors.py:113-123) on a 3-feature regression: injected noise, the draws, their unconstrained
    log-posterior and gradient, the loss, raw and clipped gradients, and the variables after one Adam(0.01) step.
    One sample is pushed to where its log density overflows float32 (dropped from the mean, vi.py:55-58)."""
    n, d, s = 40, 3, 6
    X = r32(rng.normal(size=(n, d))); w = rng.normal(size=d)
    y = r32(0.5 + X @ w + rng.normal(size=n))
    spec = cf.linreg_model(X, y)
    pu = d + 2
    params = {'loc': r32(np.concatenate([[0.4], w + 0.1, [0.3]])).astype(np.float64),
              'raw_scale': r32(rng.normal(-1.0, 0.3, pu)).astype(np.float64)}
    eps = r32(rng.normal(size=(s, pu))).astype(np.float64)
    eps[2, pu - 1] = -3000.0             # scale = softplus(-900) = 0 in float32 and float64 alike: log density not finite
    with np.errstate(all='ignore'):
        loss, grads, aux = elbo.advi_loss_and_grads(spec, params, eps, True)
        lp_u, gz = spec.log_post_unconstrained(aux['z'])
    assert not np.isfinite(lp_u[2]) and np.isfinite(np.delete(lp_u, 2)).all()
    clipped = {k_: elbo.clip_by_norm(g) for k_, g in grads.items()}
    opt = elbo.Adam(params, learning_rate=0.01)
    new = opt.step(params, clipped)
    out = os.path.join(os.path.dirname(__file__), 'advi_step_small.json')
    with open(out, 'w') as f:
        json.dump(dict(note='fp64 oracle (oracle/elbo.py) outputs on float32 inputs; not TFP outputs',
                       X=X.tolist(), y=y.tolist(), priors='b ~ Normal(0, 10), w[3] ~ Normal(0, 10), scale ~ HalfNormal(10); '
                       'bijectors Identity, Identity, Softplus',
                       loc=params['loc'].tolist(), raw_scale=params['raw_scale'].tolist(), eps=eps.tolist(),
                       z=aux['z'].tolist(), theta=aux['theta'].tolist(), finite=np.isfinite(lp_u).tolist(),
                       log_post_unconstrained=np.where(np.isfinite(lp_u), lp_u, 0.0).tolist(),
                       grad_z=np.where(np.isfinite(lp_u)[:, None], gz, 0.0).tolist(),
                       logq=np.where(np.isfinite(aux['logq']), aux['logq'], 0.0).tolist(), loss=float(loss),
                       grad={k_: v.tolist() for k_, v in grads.items()},
                       clipped={k_: v.tolist() for k_, v in clipped.items()}, learning_rate=0.01,
                       after_adam={k_: v.tolist() for k_, v in new.items()}), f)
    print('wrote', out, os.path.getsize(out), 'bytes')


if __name__ == '__main__':
    main()
