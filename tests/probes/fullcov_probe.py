"""Gradient error of the full-covariance mixture kernels against the fp64 oracle near the posterior mode,
for data centred at the origin and far from it.  Run once as is (f16 pieces) and once with BVI_FULLCOV_F32=1."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
from bayes_vi_amd import _native as nat
from oracle import closed_form as cf

dev = torch.device('cuda:0'); ws = nat.Workspace(dev)
for n, d, k, c, shift, spread in ((20000, 8, 4, 4, 0.0, 0.05), (20000, 8, 4, 4, 300.0, 0.05), (50000, 32, 8, 3, 50.0, 0.01),
                                  (50000, 32, 8, 3, 50.0, 0.001), (4000, 5, 3, 4, -1000.0, 0.05)):
    rng = np.random.default_rng(n + d + k)
    mu = rng.normal(0, 6, (k, d)) + shift
    Lt = np.tril(rng.normal(0, 0.2, (k, d, d)), -1) + np.eye(d) * rng.uniform(0.7, 1.5, (k, d, 1)) * np.eye(d)
    comp = rng.integers(0, k, n)
    Y = (mu[comp] + np.einsum('nij,nj->ni', Lt[comp], rng.normal(size=(n, d)))).astype(np.float32)
    w = rng.dirichlet(np.ones(k) * 5, c)
    th = np.concatenate([w, (mu[None] + spread * rng.normal(size=(c, k, d))).reshape(c, -1),
                         (Lt[None] * (1 + spread * rng.normal(size=(c, k, d, d)))).reshape(c, -1)], -1).astype(np.float32)
    rlp, (gw, gmu, gL) = cf.gmm_fullcov(Y, th[:, :k], th[:, k:k + k * d].reshape(c, k, d), th[:, k + k * d:].reshape(c, k, d, d))
    rg = np.concatenate([gw, gmu.reshape(c, -1), gL.reshape(c, -1)], -1)
    lp, g = nat.gmm_fullcov_logp_grad(torch.tensor(Y, device=dev), torch.tensor(th, device=dev), k, ws)
    lp = lp.cpu().numpy(); g = g.cpu().numpy()
    out = []
    for name, lo, hi in (('w', 0, k), ('mu', k, k + k * d), ('L', k + k * d, th.shape[1])):
        e = np.abs(g[:, lo:hi] - rg[:, lo:hi]).max(-1) / np.abs(rg[:, lo:hi]).max(-1)
        out.append('%s %.2e' % (name, e.max()))
    print('n=%d D=%d K=%d shift=%g spread=%g: logp rel %.2e | grad err / max|g| per block: %s'
          % (n, d, k, shift, spread, (np.abs(lp - rlp) / np.abs(rlp)).max(), '  '.join(out)), flush=True)
