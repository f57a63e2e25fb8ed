"""Gradient / log-density error of the f16-piece regression kernel and of the exact-f32 kernel against the fp64
oracle at high signal-to-noise ratios, chains at the posterior's own width around the mode (where the gradient is a
small difference of large sums)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
from bayes_vi_amd import _native as nat
from oracle import closed_form as cf

dev = torch.device('cuda:0'); ws = nat.Workspace(dev)
d, c = 64, 64
cases = [(1_000_000, s) for s in (1.0, 0.1, 0.01, 0.001)] + [(250_000, 1.0), (4_000_000, 1.0)]
for n, sigma in cases:
    rng = np.random.default_rng(17)
    X = rng.normal(size=(n, d)).astype(np.float32); w = rng.normal(size=d)
    y = (0.5 + X.astype(np.float64) @ w + sigma * rng.normal(size=n)).astype(np.float32)
    A = np.concatenate([np.ones((n, 1)), X.astype(np.float64)], 1)
    ols = np.linalg.lstsq(A, y.astype(np.float64), rcond=None)[0]
    width = sigma / np.sqrt(n)
    th = np.concatenate([ols[None, :1] + width * rng.normal(size=(c, 1)), ols[None, 1:] + width * rng.normal(size=(c, d)),
                         sigma * rng.uniform(0.9, 1.1, (c, 1))], -1).astype(np.float32)
    ref_lp, ref_g = cf.linreg(X, y, th[:16])
    Xd, yd, thd = (torch.tensor(a, device=dev) for a in (X, y, th))
    pk = nat.linreg_pack(Xd, yd, fmt=nat.PACK_F16)
    for name in ('f32', 'f16'):
        if name == 'f16':
            lp, g = nat.linreg_packed_logp_grad(pk, n, d, thd, ws, fmt=nat.PACK_F16)
        else:
            lp, g = nat.linreg_logp_grad(Xd, yd, thd, ws, precision=nat.PREC_F32)
        lp = lp.cpu().numpy()[:16].astype(np.float64); g = g.cpu().numpy()[:16].astype(np.float64)
        e_lp = np.abs(lp - ref_lp) / np.abs(ref_lp)
        gw = slice(1, 1 + d)
        e_g = np.abs(g[:, gw] - ref_g[:, gw]).max(-1) / np.abs(ref_g[:, gw]).max(-1)
        print('n=%-8d sigma=%-6g signal/noise=%-7.0f %-4s logp relerr max %.2e   weight-gradient err / max|g| max %.2e  (max|g| %.3g)'
              % (n, sigma, np.abs(X @ w).std() / sigma, name, e_lp.max(), e_g.max(), np.abs(ref_g[:, gw]).max()), flush=True)
