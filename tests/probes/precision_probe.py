"""Measures logp / gradient error of each linreg precision mode against the fp64 oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
from bayes_vi_amd import _native as nat
from oracle import closed_form as cf

dev = torch.device('cuda:0'); ws = nat.Workspace(dev)
for n, c, spread in [(64, 128, 0.05), (1000, 128, 0.05), (20000, 256, 0.05), (200000, 128, 0.01),
                     (200000, 128, 0.002), (1000000, 128, 0.01), (1000000, 128, 0.001)]:
    rng = np.random.default_rng(n + c)
    d = 64
    X = rng.normal(size=(n, d)).astype(np.float32); w = rng.normal(size=d)
    y = (0.5 + X @ w + rng.normal(size=n)).astype(np.float32)
    th = np.concatenate([0.5 + spread * rng.normal(size=(c, 1)), w + spread * rng.normal(size=(c, d)),
                         rng.uniform(0.8, 1.5, (c, 1))], -1).astype(np.float32)
    ref_lp, ref_g = cf.linreg(X, y, th[:32])
    Xd, yd, thd = (torch.tensor(a, device=dev) for a in (X, y, th))
    pkm = nat.linreg_pack(Xd, yd, fmt=nat.PACK_F16)
    for name, p in (('f32', nat.PREC_F32), ('x3', nat.PREC_BF16X3), ('x6', nat.PREC_BF16X6), ('mix', None)):
        if p is None:
            lp, g = nat.linreg_packed_logp_grad(pkm, n, d, thd, ws, fmt=nat.PACK_F16)
        else:
            lp, g = nat.linreg_logp_grad(Xd, yd, thd, ws, precision=p)
        lp = lp.cpu().numpy()[:32].astype(np.float64); g = g.cpu().numpy()[:32].astype(np.float64)
        e_lp = np.abs(lp - ref_lp) / np.abs(ref_lp)
        e_g = np.abs(g - ref_g).max(-1) / np.abs(ref_g).max(-1)
        print('n=%8d spread=%.3f %-4s logp relerr max %.2e  grad err/max|g| max %.2e  (|g|max %.1f)'
              % (n, spread, name, e_lp.max(), e_g.max(), np.abs(ref_g).max()), flush=True)
