"""Generates tests/golden/hotpath_small.json from the fp64 closed-form oracle.

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
    out = os.path.join(os.path.dirname(__file__), 'hotpath_small.json')
    with open(out, 'w') as f:
        json.dump(dict(note='closed-form fp64 oracle outputs on float32 inputs; not TFP outputs', cases=cases), f)
    print('wrote', out, os.path.getsize(out), 'bytes')


if __name__ == '__main__':
    main()
