"""Kernel time of the 1-D mixture and iid-Normal likelihoods at BASELINE.json configs[2] size (N = 100M, K = 8)."""
import os, sys, argparse
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
from bayes_vi_amd import _native as nat

ap = argparse.ArgumentParser()
ap.add_argument('--rows', type=int, default=100_000_000)
ap.add_argument('--k', type=int, default=8)
ap.add_argument('--lib', default=None)
ap.add_argument('--chains', type=int, nargs='*', default=[1, 4, 16, 64, 256, 1024])
ap.add_argument('--only', default=None)
ap.add_argument('--sorted', action='store_true', help='rows in ascending order (what the host layer stores)')
a = ap.parse_args()
if a.lib:
    nat._LIB_PATH = os.path.abspath(a.lib)
dev = torch.device('cuda:0'); ws = nat.Workspace(dev)
g = torch.Generator(device=dev); g.manual_seed(1)
k = a.k
locs = torch.linspace(-20, 20, k, device=dev)
comp = torch.randint(0, k, (a.rows,), generator=g, device=dev)
y = locs[comp] + torch.randn(a.rows, generator=g, device=dev)
del comp
if a.sorted:
    y = torch.sort(y).values.contiguous()
for c in a.chains:
    th = torch.cat([torch.full((c, k), 1.0 / k, device=dev) * (1 + 0.1 * torch.rand(c, k, generator=g, device=dev)),
                    locs + 0.1 * torch.randn(c, k, generator=g, device=dev),
                    1 + 0.1 * torch.rand(c, k, generator=g, device=dev)], -1).contiguous()
    thn = torch.cat([0.1 * torch.randn(c, 1, generator=g, device=dev), 11 + torch.rand(c, 1, generator=g, device=dev)], -1).contiguous()
    for name, fn in (('gmm1d', lambda: nat.gmm1d_logp_grad(y, th, k, ws)), ('gmm1d-value', lambda: nat.gmm1d_logp_grad(y, th, k, ws, want_grad=False)),
                     ('normal', lambda: nat.normal_iid_logp_grad(y, thn, ws))):
        if a.only and not name.startswith(a.only):
            continue
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ts.sort(); med = ts[2]
        # calls shorter than ~0.3 ms: one call between two events also counts the host's share of issuing it (output
        # tensors, the ctypes call, two launches: 30 - 40 us during which the GPU waits); eight calls queued back to
        # back give the GPU's own time per call
        note = ''
        if med < 0.3:
            bs = []
            for _ in range(5):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _q in range(8):
                    fn()
                e1.record(); torch.cuda.synchronize(); bs.append(e0.elapsed_time(e1) / 8)
            bs.sort()
            note = '  (back to back: %.3f ms, %.1f GB/s alg)' % (bs[2], 4.0 * a.rows / bs[2] / 1e6)
        print('%-11s rows=%d K=%d chains=%4d: %8.3f ms  %7.1f GB/s alg  %8.2f G row-chain-evals/s%s'
              % (name, a.rows, k, c, med, 4.0 * a.rows / med / 1e6, a.rows * c / med / 1e6, note), flush=True)
