"""Structure of the f16-piece regression kernel's gradient error at the posterior mode (N = 1M, D = 64, chains within one
posterior standard deviation): per chain / per feature, against the exact-f32 kernel, and regressed on candidate
directions.  Finding: the error is the same for every chain, sits on the features with the smallest weights and has
the sign of that weight -- sign(w_d) * const * sum_i |x_id| -- which is what cutting off small addends toward zero
inside the matrix instruction produces (see mfma_rounding.hip).  That was with a floating hi / lo split of W; with the
first piece on a grid relative to the chain's largest weight (first_piece in linreg_f16.hip) the feature-specific part
is gone."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
from bayes_vi_amd import _native as nat
from oracle import closed_form as cf
dev = torch.device('cuda:0'); ws = nat.Workspace(dev)
n, d, c, sigma = 1_000_000, 64, 64, 1.0
rng = np.random.default_rng(17)
X = rng.normal(size=(n, d)).astype(np.float32); w = rng.normal(size=d)
y = (0.5 + X.astype(np.float64) @ w + sigma * rng.normal(size=n)).astype(np.float32)
A = np.concatenate([np.ones((n, 1)), X.astype(np.float64)], 1)
ols = np.linalg.lstsq(A, y.astype(np.float64), rcond=None)[0]
width = sigma / np.sqrt(n)
th = np.concatenate([ols[None, :1] + width * rng.normal(size=(c, 1)), ols[None, 1:] + width * rng.normal(size=(c, d)),
                     sigma * np.ones((c, 1))], -1).astype(np.float32)
ref_lp, ref_g = cf.linreg(X, y, th[:16])
Xd, yd, thd = (torch.tensor(a, device=dev) for a in (X, y, th))
pk = nat.linreg_pack(Xd, yd, fmt=nat.PACK_F16)
lp, g = nat.linreg_packed_logp_grad(pk, n, d, thd, ws, fmt=nat.PACK_F16)
g = g.cpu().numpy()[:16].astype(np.float64)
lp32, g32 = nat.linreg_logp_grad(Xd, yd, thd, ws, precision=nat.PREC_F32)
g32 = g32.cpu().numpy()[:16].astype(np.float64)
X64 = X.astype(np.float64)
M = X64.T @ X64
E = (g[:, 1:1 + d] - ref_g[:, 1:1 + d]); E32 = (g32[:, 1:1 + d] - ref_g[:, 1:1 + d])
print('per-chain max|e| f16:', np.round(np.abs(E).max(1), 4))
print('per-chain argmax feature :', np.abs(E).argmax(1))
print('per-chain max|e| f32:', np.round(np.abs(E32).max(1), 4))
print('per-feature rms e f16 (first 16):', np.round(np.sqrt((E ** 2).mean(0))[:16], 4))
print('mean over chains of e (first 16):', np.round(E.mean(0)[:16], 4))
print('rms of chain-mean e %.4g vs rms of e %.4g' % (np.sqrt((E.mean(0) ** 2).mean()), np.sqrt((E ** 2).mean())))
for ch in range(2):
    e = (g[ch, 1:1 + d] - ref_g[ch, 1:1 + d]) * sigma ** 2
    wv = th[ch, 1:1 + d].astype(np.float64)
    cands = {'M w': M @ wv, 'colsum': X64.sum(0), 'sign(w) sum|x|': np.sign(wv) * np.abs(X64).sum(0), 'N w': n * wv,
             'X^T y': X64.T @ y.astype(np.float64)}
    print('chain', ch, '|e| rms %.3g, e_b0 %.3g' % (np.sqrt((e ** 2).mean()), (g[ch, 0] - ref_g[ch, 0])))
    for k, v in cands.items():
        coef = (e @ v) / (v @ v)
        corr = (e @ v) / np.sqrt((e @ e) * (v @ v))
        print('   %-16s corr %+.3f  coef %+.3e' % (k, corr, coef))
em = E.mean(0)
wv = th[0, 1:1 + d].astype(np.float64)
order = np.argsort(-np.abs(em))[:8]
print('worst features', order, 'e', np.round(em[order], 4))
print('their w      ', np.round(wv[order], 3))
print('their max|x| ', np.round(np.abs(X64).max(0)[order], 3))
print('largest |w| features', np.argsort(-np.abs(wv))[:8], np.round(np.sort(-np.abs(wv))[:8], 3))
for name, v in (('w', wv), ('|w|', np.abs(wv)), ('w^3', wv ** 3), ('frac part of log2|w|', np.log2(np.abs(wv)) % 1)):
    print('corr(e, %s) = %+.3f' % (name, np.corrcoef(em, v)[0, 1]))
