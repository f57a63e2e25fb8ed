"""Times the full-covariance mixture entry point at the BASELINE.json configs[3] shape (N, C chosen here)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
from bayes_vi_amd import _native as nat
if '--lib' in sys.argv:
    nat._LIB_PATH = os.path.abspath(sys.argv[sys.argv.index('--lib') + 1])
n, d, k, c = 1_000_000, 32, 16, 64
dev = torch.device('cuda:0'); ws = nat.Workspace(dev)
g = torch.Generator(device=dev); g.manual_seed(5)
mu = torch.randint(-20, 21, (k, d), generator=g, device=dev).float()
comp = torch.randint(0, k, (n,), generator=g, device=dev)
Y = mu[comp] + torch.randn(n, d, generator=g, device=dev)
L = torch.tril(torch.randn(c, k, d, d, generator=g, device=dev) * 0.05)
L = L + torch.diag_embed(1.0 + 0.1 * torch.rand(c, k, d, generator=g, device=dev))
if '--clustered' in sys.argv:      # the order the host layer stores the rows in (GaussianMixtureFullCov)
    from bayes_vi_amd.fused_target import cluster_centres
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    nat.cluster_rows(Y[:100000].contiguous(), cluster_centres(k)); torch.cuda.synchronize()          # loads the kernels
    e0.record(); Y = nat.cluster_rows(Y, cluster_centres(k)); e1.record(); torch.cuda.synchronize()
    print('rows stored cluster by cluster (once per dataset): %.1f ms' % e0.elapsed_time(e1))
w = torch.rand(c, k, generator=g, device=dev) + 0.5
th = torch.cat([w, (mu[None] + 0.1 * torch.randn(c, k, d, generator=g, device=dev)).reshape(c, -1), L.reshape(c, -1)], -1).contiguous()
for _ in range(2):
    lp, gr = nat.gmm_fullcov_logp_grad(Y, th, k, ws)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); lp, gr = nat.gmm_fullcov_logp_grad(Y, th, k, ws); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
fl = 6.0 * n * c * k * d * d
print('fullcov N=%d D=%d K=%d C=%d: median %.2f ms -> %.1f TFLOP/s (6 N C K D^2), %.1f K grad-evals/s, finite=%s'
      % (n, d, k, c, ts[2], fl / ts[2] / 1e9, c / ts[2], bool(torch.isfinite(lp).all())))
for _ in range(2):
    nat.gmm_fullcov_logp_grad(Y, th, k, ws, want_grad=False)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); lpv, _ = nat.gmm_fullcov_logp_grad(Y, th, k, ws, want_grad=False); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
print('fullcov N=%d D=%d K=%d C=%d value only: median %.2f ms, same log-densities: %s' % (n, d, k, c, ts[2], bool(torch.equal(lp, lpv))))
