"""Times the linreg entry point at the headline shape and checks it against the exact-f32 kernel."""
import os, sys, argparse
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
from bayes_vi_amd import _native as nat

ap = argparse.ArgumentParser()
ap.add_argument('--rows', type=int, default=10_000_000)
ap.add_argument('--chains', type=int, default=1024)
ap.add_argument('--d', type=int, default=64)
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--prec', default='bf16x6')
ap.add_argument('--check', action='store_true')
ap.add_argument('--packed', action='store_true')
ap.add_argument('--f16', action='store_true')
ap.add_argument('--value-only', action='store_true', help='no gradient (what random-walk Metropolis calls)')
ap.add_argument('--lib', default=None, help='alternative build of the library (kernel experiments)')
a = ap.parse_args()
if a.lib:
    nat._LIB_PATH = os.path.abspath(a.lib)
dev = torch.device('cuda:0'); ws = nat.Workspace(dev)
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.empty(a.rows, a.d, device=dev)
for s in range(0, a.rows, 1 << 20):
    X[s:s + (1 << 20)].normal_(generator=g)
w = torch.randn(a.d, generator=g, device=dev)
y = 0.5 + X @ w + torch.randn(a.rows, generator=g, device=dev)
th = torch.cat([0.5 + 0.01 * torch.randn(a.chains, 1, generator=g, device=dev),
                w + 0.01 * torch.randn(a.chains, a.d, generator=g, device=dev),
                1 + 0.01 * torch.rand(a.chains, 1, generator=g, device=dev)], -1)
prec = {'auto': 0, 'f32': 1, 'bf16x3': 2, 'bf16x6': 3}[a.prec]
if a.f16:
    pk = nat.linreg_pack(X, y, fmt=nat.PACK_F16)
    run = lambda: nat.linreg_packed_logp_grad(pk, a.rows, a.d, th, ws, precision=0, fmt=nat.PACK_F16,
                                              want_grad=not a.value_only)
elif a.packed:
    pk = nat.linreg_pack(X, y)
    run = lambda: nat.linreg_packed_logp_grad(pk, a.rows, a.d, th, ws, precision=prec)
else:
    run = lambda: nat.linreg_logp_grad(X, y, th, ws, precision=prec)
for _ in range(2):
    lp, gr = run()
torch.cuda.synchronize()
ts = []
for _ in range(a.reps):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); lp, gr = run(); e1.record()
    torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ts.sort()
med = ts[len(ts) // 2]
fl = 4.0 * a.chains * a.rows * a.d
label = 'f16 plane (auto)' if a.f16 else ('%s, packed bf16 pieces' % a.prec if a.packed else a.prec)
print('kernel=%s%s rows=%d chains=%d: median %.3f ms  min %.3f  -> %.1f alg TFLOP/s, %.1f K grad-evals/s'
      % (label, ' value-only' if a.f16 and a.value_only else '', a.rows, a.chains, med, ts[0], fl / med / 1e9,
         a.chains / med), flush=True)
if a.check:
    lp2, gr2 = nat.linreg_logp_grad(X, y, th[:64].contiguous(), ws, precision=1)
    e1 = ((lp[:64] - lp2).abs() / lp2.abs()).max().item()
    e2 = ((gr[:64] - gr2).abs().max(-1).values / gr2.abs().max(-1).values).max().item()
    print('vs exact f32 kernel on 64 chains: logp rel %.2e  grad err/max|g| %.2e' % (e1, e2), flush=True)
