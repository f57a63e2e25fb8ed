"""Device HMC on a regression beyond the one-workgroup-per-chain kernel: per-step launches against the
transition replayed as a HIP graph.  Usage: hmc_graph_bench.py [N] [D] [C] [L]"""
import collections, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bayesian-inference_amd'))
import bayes_vi_amd.distributions as tfd
from bayes_vi_amd import symbolic
from bayes_vi_amd.model import Model
from bayes_vi_amd.inference.mcmc import MCMC
from bayes_vi_amd.inference.mcmc.device_hmc import DeviceHamiltonianMonteCarloKernel
from bayes_vi_amd.utils.datasets import TensorDataset

n, d, c, L = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (200_000, 32, 64, 10)
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn(n, d, device=dev, generator=g)
w = torch.randn(d, device=dev, generator=g)
y = X @ w + 0.3 + torch.randn(n, device=dev, generator=g)
priors = collections.OrderedDict(b=tfd.Normal(0., 10.), w=tfd.Normal(torch.zeros(d), 10.), scale=tfd.HalfNormal(10.))
model = Model(priors, lambda b, w, scale, features: tfd.Normal(loc=b + symbolic.matvec(features, w), scale=scale))
post = MCMC(model, TensorDataset(X, y)).posterior
z0 = torch.cat([torch.full((c, 1), 0.3), w.cpu().repeat(c, 1), torch.full((c, 1), 0.55)], -1).to(dev)
z0 = z0 + 0.3 / np.sqrt(n) * torch.randn(z0.shape, device=dev, generator=g)
eps = 0.35 / np.sqrt(n)
T = 200
for name in ('launch per step', 'many_steps', 'graph replay') * 3:
    kern = DeviceHamiltonianMonteCarloKernel(post, eps, L, seed=1, graph_replay=name[0] == 'g')
    z, st = z0.clone(), None
    st = kern.bootstrap(z)
    z, st, _ = kern.one_step(z, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if name[0] == 'l':
        for _ in range(T):
            z, st, tr = kern.one_step(z, st)
        acc = float(tr['is_accepted'].float().mean())
    else:
        zs, z, st, tr = kern.many_steps(z, st, T)
        acc = float(tr['is_accepted'].float().mean())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('%-16s N=%d D=%d C=%d L=%d: %.3f ms/transition, %.1f us/leapfrog, accept %.2f'
          % (name, n, d, c, L, 1e3 * dt / T, 1e6 * dt / T / L, acc), flush=True)
