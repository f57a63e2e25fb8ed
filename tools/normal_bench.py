"""Kernel time of the iid-Normal likelihood at N = 100M rows, with and without the gradient."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
from bayes_vi_amd import _native as nat
dev = torch.device('cuda:0'); ws = nat.Workspace(dev)
y = torch.randn(100_000_000, device=dev) * 3 + 7
for c in (1, 16, 64, 1024):
    th = torch.stack([7 + 0.01 * torch.randn(c, device=dev), 3 + 0.1 * torch.rand(c, device=dev)], -1).contiguous()
    for wg in (True, False):
        for _ in range(3):
            nat.normal_iid_logp_grad(y, th, ws, want_grad=wg)
        torch.cuda.synchronize(); ts = []
        for _ in range(7):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); nat.normal_iid_logp_grad(y, th, ws, want_grad=wg); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print('normal 100M rows, %4d chains, %s: %.3f ms' % (c, 'value + gradient' if wg else 'value only', sorted(ts)[3]), flush=True)
