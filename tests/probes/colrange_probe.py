"""Per-column gradient error of the f16-piece kernel and the exact-f32 kernel on badly scaled columns."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
from bayes_vi_amd import _native as nat
from oracle import closed_form as cf

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
X64, y64, th64 = X.astype(np.float64), y.astype(np.float64), th.astype(np.float64)
r = np.abs(y64[None] - th64[:, :1] - th64[:, 1:1 + d] @ X64.T)
s2 = th64[:, -1:] ** 2
ts = np.concatenate([r.sum(-1, keepdims=True), r @ np.abs(X64),
                     (r * r).sum(-1, keepdims=True) / th64[:, -1:] + n * th64[:, -1:]], -1) / s2
dev = 'cuda:0'
Xd, yd, thd = [torch.tensor(a, device=dev) for a in (X, y, th)]
ws = nat.Workspace(torch.device(dev))
pk = nat.linreg_pack(Xd, yd, fmt=nat.PACK_F16)
out = {'f16': nat.linreg_packed_logp_grad(pk, n, d, thd, ws, fmt=nat.PACK_F16),
       'f32': nat.linreg_logp_grad(Xd, yd, thd, ws, precision=nat.PREC_F32),
       'x6': nat.linreg_logp_grad(Xd, yd, thd, ws, precision=nat.PREC_BF16X6)}
print('typical |r|', np.median(r), ' |z| ', np.median(np.abs(th64[:, 1:1 + d] @ X64.T)))
for name, (lp, g) in out.items():
    g = g.cpu().numpy(); lp = lp.cpu().numpy()
    err = np.abs(g - ref_g)
    rel = err / (1e-5 * np.abs(ref_g) + 3e-7 * ts + 1e-30)
    print(name, 'logp rel', np.abs(lp - ref_lp).max() / np.abs(ref_lp).max(), ' worst ratio to budget', rel.max(),
          'at col', np.unravel_index(rel.argmax(), rel.shape), ' err/ts max', (err / (ts + 1e-30)).max())
    print('   per-col max ratio:', np.round(rel.max(0)[:12], 2), '... b0', round(rel.max(0)[0], 2), 'sigma', round(rel.max(0)[-1], 2))
