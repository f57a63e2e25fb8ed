"""Which entries of the column-range test case are outside the bar (f16-plane kernel vs fp64 oracle)."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo/bayesian-inference_amd'); sys.path.insert(0, '/root/repo')
from bayes_vi_amd import _native as nat
from oracle import closed_form as cf
dev = torch.device('cuda:0'); ws = nat.Workspace(dev)
rng = np.random.default_rng(123)
n, d, c = int(sys.argv[1]) if len(sys.argv) > 1 else 30000, 64, 160
colscale = 10.0 ** rng.uniform(-3, 3, d)
X = rng.normal(size=(n, d)) + rng.normal(size=d) * 2.0
X[:, 5] = np.exp(1.5 * rng.normal(size=n)); X[:, 6] = 1.0
X = X * colscale; X[:, 7] = 0.0; X = X.astype(np.float32)
w = rng.normal(size=d) / colscale
y = (300.0 + X.astype(np.float64) @ w + 3.0 * rng.normal(size=n)).astype(np.float32)
th = np.concatenate([300.0 + 0.3 * rng.normal(size=(c, 1)), w * (1 + 0.01 * rng.normal(size=(c, d))),
                     rng.uniform(2.0, 4.0, (c, 1))], -1).astype(np.float32)
ref_lp, ref_g = cf.linreg(X, y, th)
Xd, yd, thd = [torch.tensor(a, device=dev) for a in (X, y, th)]
pk = nat.linreg_pack(Xd, yd, fmt=nat.PACK_F16)
lp, g = nat.linreg_packed_logp_grad(pk, n, d, thd, ws, fmt=nat.PACK_F16)
lp32, g32 = nat.linreg_logp_grad(Xd, yd, thd, ws, precision=nat.PREC_F32)
g = g.cpu().numpy(); g32 = g32.cpu().numpy()
b0, W, sc = th[:, 0].astype(np.float64), th[:, 1:1 + d].astype(np.float64), th[:, d + 1].astype(np.float64)
r = y[None] - b0[:, None] - W @ X.T.astype(np.float64)
ts = np.concatenate([np.abs(r).sum(-1)[:, None], np.abs(r) @ np.abs(X.astype(np.float64)), (r * r).sum(-1)[:, None] / sc[:, None] + n * sc[:, None]], -1) / sc[:, None] ** 2
for name, gg in (('f16', g), ('f32', g32)):
    err = np.abs(gg - ref_g); tol = 1e-5 * np.abs(ref_g) + 3e-7 * ts + 1e-30
    ratio = err / tol
    print(name, 'logp rel', (np.abs((lp if name == 'f16' else lp32).cpu().numpy() - ref_lp) / np.abs(ref_lp)).max(),
          'worst err/tol per column:', np.round(ratio.max(0), 2))
