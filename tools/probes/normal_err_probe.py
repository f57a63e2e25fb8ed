import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
from bayes_vi_amd import _native as nat
dev = torch.device('cuda:0'); ws = nat.Workspace(dev)
g = torch.Generator(device=dev); g.manual_seed(5)
n = 100_000_000
y = torch.empty(n, device=dev).normal_(7.0, 3.0, generator=g)
c = 64
for name, mus in (('random mu', 7 + 3e-4 * torch.randn(c, generator=g, device=dev)),
                  ('mu on a 2^-12 grid', 7 + torch.round(3e-4 * torch.randn(c, generator=g, device=dev) * 4096) / 4096)):
    th = torch.stack([mus, 3 * (1 + 1e-4 * torch.randn(c, generator=g, device=dev))], -1).contiguous()
    mu, sc = th[:, 0].double(), th[:, 1].double()
    s2 = torch.zeros(c, dtype=torch.float64, device=dev); s2f = torch.zeros(c, dtype=torch.float64, device=dev)
    for s in range(0, n, 1 << 22):
        r = y[s:s + (1 << 22)].double()[None] - mu[:, None]
        s2 += (r * r).sum(-1)
        rf = (y[s:s + (1 << 22)][None] - th[:, :1]).double()      # the float subtraction the kernel does
        s2f += (rf * rf).sum(-1)
    ref64 = -0.5 * s2 / sc ** 2 - n * torch.log(sc) - 0.5 * n * math.log(2 * math.pi)
    level = torch.tensor([float(ref64[0])], dtype=torch.float64, device=dev)
    lp, _ = nat.normal_iid_logp_grad(y, th, ws, logp_ref=level)
    err = lp.double() + level - ref64
    print(name, 'err', [round(v, 4) for v in err[:8].tolist()], 'rms', float(err.pow(2).mean().sqrt()))
    print('   effect of rounding y - mu to float on lp:', [round(v, 4) for v in (-0.5 * (s2f - s2) / sc ** 2)[:8].tolist()], flush=True)
