"""Time per NUTS transition on a regression too large for the one-workgroup-per-chain kernel:
host tree bookkeeping in torch ops (samplers.NoUTurnSamplerKernel) against the device bookkeeping
(BatchedDeviceNoUTurnSamplerKernel), same posterior, same step size.  Usage: nuts_bench.py [N] [D] [C]"""
import collections
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bayesian-inference_amd'))
import bayes_vi_amd.distributions as tfd
from bayes_vi_amd import symbolic
from bayes_vi_amd.model import Model
from bayes_vi_amd.inference.mcmc import MCMC, samplers
from bayes_vi_amd.utils.datasets import TensorDataset
from bayes_vi_amd.inference.mcmc.device_hmc import BatchedDeviceNoUTurnSamplerKernel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
c = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn(n, d, device=dev, generator=g)
w = torch.randn(d, device=dev, generator=g)
y = X @ w + 0.3 + torch.randn(n, device=dev, generator=g)
priors = collections.OrderedDict(b=tfd.Normal(0., 10.), w=tfd.Normal(torch.zeros(d), 10.),
                                 scale=tfd.HalfNormal(10.))
model = Model(priors, lambda b, w, scale, features: tfd.Normal(loc=b + symbolic.matvec(features, w),
                                                                       scale=scale))
ds = TensorDataset(X, y)
post = MCMC(model, ds).posterior
z0 = torch.cat([torch.full((c, 1), 0.3), w.cpu().repeat(c, 1), torch.full((c, 1), 0.55)], -1).to(dev)
z0 = z0 + 0.3 / np.sqrt(n) * torch.randn(z0.shape, device=dev, generator=g)
eps = 0.35 / np.sqrt(n)
T = 20
for name, kern in (('host bookkeeping', samplers.NoUTurnSamplerKernel(post.unconstrained_log_prob_and_grad, eps, 5)),
                   ('device bookkeeping', BatchedDeviceNoUTurnSamplerKernel(post, eps, 5, seed=1))):
    z, st = z0.clone(), None
    st = kern.bootstrap(z)
    gen = torch.Generator(device=dev).manual_seed(1)
    for _ in range(3):
        z, st, tr = kern.one_step(z, st, gen) if name[0] == 'h' else kern.one_step(z, st)
    torch.cuda.synchronize(); t0 = time.perf_counter(); leaps = 0
    for _ in range(T):
        z, st, tr = kern.one_step(z, st, gen) if name[0] == 'h' else kern.one_step(z, st)
        leaps += int(tr['leapfrogs_taken'].max())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('%-20s N=%d D=%d C=%d: %.2f ms/transition, %.1f lock-step leapfrogs/transition, %.3f ms/leapfrog, accept %.2f'
          % (name, n, d, c, 1e3 * dt / T, leaps / T, 1e3 * dt / max(leaps, 1),
             float(torch.exp(tr['log_accept_ratio']).mean())), flush=True)
