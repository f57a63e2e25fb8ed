"""The six-decade column-range case of tests/test_gpu_parity.py with the worst error / allowance ratio per column:
used to choose the grid of the first W piece (kWGrid in linreg_f16.hip)."""
import os, sys
import numpy as np, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from bayes_vi_amd import _native as nat
from oracle import closed_form as cf
dev = torch.device('cuda:0'); ws = nat.Workspace(dev)
rng = np.random.default_rng(123)
n, d, c = 30000, 64, 160
colscale = 10.0 ** rng.uniform(-3, 3, d)
X = rng.normal(size=(n, d)) + rng.normal(size=d) * 2.0
X[:, 5] = np.exp(1.5 * rng.normal(size=n))
X[:, 6] = 1.0
X = X * colscale
X[:, 7] = 0.0
X = X.astype(np.float32)
w = rng.normal(size=d) / colscale
y = (300.0 + X.astype(np.float64) @ w + 3.0 * rng.normal(size=n)).astype(np.float32)
th = np.concatenate([300.0 + 0.3 * rng.normal(size=(c, 1)), w * (1 + 0.01 * rng.normal(size=(c, d))),
                     rng.uniform(2.0, 4.0, (c, 1))], -1).astype(np.float32)
ref_lp, ref_g = cf.linreg(X, y, th)
Xd, yd, thd = (torch.tensor(a, device=dev) for a in (X, y, th))
pk = nat.linreg_pack(Xd, yd, fmt=nat.PACK_F16)
lp, g = nat.linreg_packed_logp_grad(pk, n, d, thd, ws, fmt=nat.PACK_F16)
g = g.cpu().numpy().astype(np.float64)
X64, y64, th64 = X.astype(np.float64), y.astype(np.float64), th.astype(np.float64)
r = np.abs(y64[None] - th64[:, :1] - th64[:, 1:1 + d] @ X64.T)
s2 = th64[:, -1:] ** 2
ts = np.concatenate([r.sum(-1, keepdims=True), r @ np.abs(X64), (r * r).sum(-1, keepdims=True) / th64[:, -1:] + n * th64[:, -1:]], -1) / s2
err = np.abs(g - ref_g)
allow = 1e-5 * np.abs(ref_g) + 3e-7 * ts + 1e-30
ratio = err / allow
print('worst ratio per column (top 10):', np.argsort(-ratio.max(0))[:10], np.round(np.sort(-ratio.max(0))[:10], 2))
print('fraction of entries over:', (ratio > 1).mean())
sw = (th64[0, 1:1 + d] * np.abs(X64).max(0))
print('|w_d| * max|x_d| (first 12):', np.round(sw[:12], 2), ' max', sw.max())
