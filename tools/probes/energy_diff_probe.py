"""Error of log-density DIFFERENCES between nearby parameter vectors at full size (what a Metropolis test uses),
with the kernels' outputs taken relative to a level (logp_ref): regression N = 10M, mixture and Normal N = 100M."""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from bayes_vi_amd import _native as nat
dev = torch.device('cuda:0'); ws = nat.Workspace(dev)
g = torch.Generator(device=dev); g.manual_seed(5)


def report(name, rel, ref, ref64):
    err = (rel.double() + ref) - ref64
    d = err - err[0]
    print('%-34s |lp| %.3g  abs err of lp: mean %.3g  spread (differences) max %.3g  rms %.3g'
          % (name, float(ref64.abs().max()), float(err.mean()), float(d.abs().max()), float(d.pow(2).mean().sqrt())), flush=True)


def level_of(lp):
    ref = torch.zeros(1, dtype=torch.float64, device=dev)
    nat.logp_rebase(lp.clone(), 1, ref)
    return ref


# ---- regression 10M x 64
n, d, c = 10_000_000, 64, 64
X = torch.empty(n, d, device=dev)
for s in range(0, n, 1 << 20):
    X[s:s + (1 << 20)].normal_(generator=g)
w = torch.randn(d, generator=g, device=dev)
y = 0.5 + X @ w + torch.randn(n, generator=g, device=dev)
A = torch.zeros(d + 1, d + 1, dtype=torch.float64, device=dev); b = torch.zeros(d + 1, dtype=torch.float64, device=dev)
yy = torch.zeros((), dtype=torch.float64, device=dev)
for s in range(0, n, 1 << 20):
    Xa = torch.cat([torch.ones(min(1 << 20, n - s), 1, dtype=torch.float64, device=dev), X[s:s + (1 << 20)].double()], 1)
    yc = y[s:s + (1 << 20)].double()
    A += Xa.T @ Xa; b += Xa.T @ yc; yy += yc @ yc
beta = torch.linalg.solve(A, b); sig = torch.sqrt((yy - beta @ b) / n)
th = torch.cat([beta[None] + 3e-4 * torch.randn(c, d + 1, generator=g, device=dev, dtype=torch.float64),
                sig * (1 + 2e-4 * torch.randn(c, 1, generator=g, device=dev, dtype=torch.float64))], -1).float().contiguous()
t64 = th.double(); bb, sc = t64[:, :d + 1], t64[:, d + 1]
s2 = yy - 2 * bb @ b + ((bb @ A) * bb).sum(-1)
ref64 = -0.5 * s2 / sc ** 2 - n * torch.log(sc) - 0.5 * n * math.log(2 * math.pi)
pk = nat.linreg_pack(X, y, fmt=nat.PACK_F16)
lp_abs, _ = nat.linreg_packed_logp_grad(pk, n, d, th, ws, fmt=nat.PACK_F16)
ref = level_of(lp_abs)
report('regression f16 plane, absolute', lp_abs, torch.zeros((), dtype=torch.float64, device=dev), ref64)
lp, _ = nat.linreg_packed_logp_grad(pk, n, d, th, ws, fmt=nat.PACK_F16, logp_ref=ref)
report('regression f16 plane, 64 chains', lp, ref, ref64)
lp, _ = nat.linreg_packed_logp_grad(pk, n, d, th[:16].contiguous(), ws, fmt=nat.PACK_F16, logp_ref=ref)
report('regression f16 plane, 16 chains', lp, ref, ref64[:16])
lp, _ = nat.linreg_logp_grad(X, y, th[:16].contiguous(), ws, precision=nat.PREC_F32, logp_ref=ref)
report('regression exact f32, 16 chains', lp, ref, ref64[:16])
del X, A, pk

# ---- Normal 100M
n = 100_000_000
y = torch.empty(n, device=dev).normal_(7.0, 3.0, generator=g)
s0 = torch.zeros((), dtype=torch.float64, device=dev); s1 = s0.clone(); s2 = s0.clone()
for s in range(0, n, 1 << 24):
    yc = y[s:s + (1 << 24)].double(); s1 += yc.sum(); s2 += (yc * yc).sum()
mu_hat = s1 / n; sd_hat = torch.sqrt(s2 / n - mu_hat ** 2)
th = torch.stack([mu_hat + 3e-4 * torch.randn(c, generator=g, device=dev, dtype=torch.float64),
                  sd_hat * (1 + 1e-4 * torch.randn(c, generator=g, device=dev, dtype=torch.float64))], -1).float().contiguous()
mu, sc = th[:, 0].double(), th[:, 1].double()
ref64 = -0.5 * (s2 - 2 * mu * s1 + n * mu * mu) / sc ** 2 - n * torch.log(sc) - 0.5 * n * math.log(2 * math.pi)
lp_abs, _ = nat.normal_iid_logp_grad(y, th, ws)
ref = level_of(lp_abs)
lp, _ = nat.normal_iid_logp_grad(y, th, ws, logp_ref=ref)
report('Normal 100M rows, 64 chains', lp, ref, ref64)

# ---- mixture 100M
k = 8
locs = torch.linspace(-20, 20, k, device=dev)
for s in range(0, n, 1 << 24):
    m = min(1 << 24, n - s)
    y[s:s + m] = locs[torch.randint(0, k, (m,), generator=g, device=dev)] + torch.randn(m, generator=g, device=dev)
c = 64
th = torch.cat([(1.0 / k) * (1 + 1e-4 * torch.randn(c, k, generator=g, device=dev)),
                locs + 1e-4 * torch.randn(c, k, generator=g, device=dev),
                1 + 1e-4 * torch.randn(c, k, generator=g, device=dev)], -1).contiguous()
from test_gpu_fullsize import _mixture_float64
sel = torch.arange(8, device=dev)
ref64, _, _ = _mixture_float64(y, th[sel], k)
lp_abs, _ = nat.gmm1d_logp_grad(y, th, k, ws)
ref = level_of(lp_abs)
lp, _ = nat.gmm1d_logp_grad(y, th, k, ws, logp_ref=ref)
report('mixture 100M rows (with gradient)', lp[sel], ref, ref64)
lp, _ = nat.gmm1d_logp_grad(y, th, k, ws, want_grad=False, logp_ref=ref)
report('mixture 100M rows (value only)', lp[sel], ref, ref64)
