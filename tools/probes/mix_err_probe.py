import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from bayes_vi_amd import _native as nat
from test_gpu_fullsize import _mixture_float64
dev = torch.device('cuda:0'); ws = nat.Workspace(dev)
g = torch.Generator(device=dev); g.manual_seed(5)
n, k, c = 100_000_000, 8, 64
locs = torch.linspace(-20, 20, k, device=dev)
y = torch.empty(n, device=dev)
for s in range(0, n, 1 << 24):
    m = min(1 << 24, n - s)
    y[s:s + m] = locs[torch.randint(0, k, (m,), generator=g, device=dev)] + torch.randn(m, generator=g, device=dev)
for name, grid in (('random locs', 0), ('locs on a 2^-10 grid', 1024), ('locs and scales on a grid', -1024)):
    mu = locs + 1e-4 * torch.randn(c, k, generator=g, device=dev)
    sc = 1 + 1e-4 * torch.randn(c, k, generator=g, device=dev)
    if grid:
        mu = torch.round(mu * abs(grid)) / abs(grid)
    if grid < 0:
        sc = torch.round(sc * 4096) / 4096
    th = torch.cat([(1.0 / k) * (1 + 1e-4 * torch.randn(c, k, generator=g, device=dev)), mu, sc], -1).contiguous()
    sel = torch.arange(8, device=dev)
    ref64, _, _ = _mixture_float64(y, th[sel], k)
    level = torch.tensor([float(ref64[0])], dtype=torch.float64, device=dev)
    for wg in (True, False):
        lp, _ = nat.gmm1d_logp_grad(y, th, k, ws, want_grad=wg, logp_ref=level)
        err = lp[sel].double() + level - ref64
        print(name, 'grad' if wg else 'value', [round(v, 4) for v in err.tolist()], flush=True)
    lp, _ = nat.gmm1d_logp_grad(y, th[:4].contiguous(), k, ws, logp_ref=level)
    print(name, '4 chains', [round(v, 4) for v in (lp.double() + level - ref64[:4]).tolist()], flush=True)
