"""Full-covariance mixture on clustered, packed rows as the clusters move from well separated to overlapping:
N = 1M, D = 32, K = 16, 64 chains; centres drawn with standard deviation `spread` per coordinate (unit-variance
clusters).  Prints the time per evaluation with gradient and the share of (tile, chain) pairs by number of components
the bounds leave."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
from bayes_vi_amd import _native as nat
if '--lib' in sys.argv:
    nat._LIB_PATH = os.path.abspath(sys.argv[sys.argv.index('--lib') + 1])
n, d, k, c = 1_000_000, 32, 16, 64
dev = torch.device('cuda:0'); ws = nat.Workspace(dev)
spreads = [float(x) for x in sys.argv[sys.argv.index('--spreads') + 1].split(',')] if '--spreads' in sys.argv else [8.0, 4.0, 2.0, 1.2, 0.8, 0.4]
for spread in spreads:
    g = torch.Generator(device=dev); g.manual_seed(int(10 * spread))
    centers = spread * torch.randn(k, d, generator=g, device=dev)
    Y = (centers[torch.randint(0, k, (n,), generator=g, device=dev)] + torch.randn(n, d, generator=g, device=dev)).contiguous()
    L = torch.eye(d, device=dev)[None, None] * (1 + 0.1 * torch.rand(c, k, 1, 1, generator=g, device=dev)) \
        + 0.02 * torch.tril(torch.randn(c, k, d, d, generator=g, device=dev), -1)
    th = torch.cat([(1.0 / k) * (1 + 0.1 * torch.rand(c, k, generator=g, device=dev)),
                    (centers[None] + 0.05 * torch.randn(c, k, d, generator=g, device=dev)).reshape(c, -1),
                    L.reshape(c, -1)], -1).contiguous()
    out = []
    for label, rows in (('as drawn', Y), ('clustered', nat.cluster_rows(Y, 64))):
        packed = nat.gmm_fullcov_pack(rows)
        for _ in range(2):
            nat.gmm_fullcov_logp_grad(rows, th, k, ws, packed=packed)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); nat.gmm_fullcov_logp_grad(rows, th, k, ws, packed=packed); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        out.append('%s %.2f ms' % (label, ts[2]))
    print('centres spread %.1f: %s' % (spread, ', '.join(out)), flush=True)
