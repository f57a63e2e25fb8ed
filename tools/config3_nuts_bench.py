"""BASELINE.json configs[3] shape (full-covariance mixture, D = 32, K = 16, NUTS): time per lock-step leapfrog
of (a) host tree bookkeeping + posterior assembly in torch ops, (b) host bookkeeping + device assembly,
(c) device bookkeeping + device assembly.  Usage: config3_nuts_bench.py [N] [C]  (N = rows on this GPU)."""
import collections, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
import bayes_vi_amd.distributions as tfd
import bayes_vi_amd.bijectors as tfb
from bayes_vi_amd.model import Model
from bayes_vi_amd.inference.mcmc import MCMC, samplers
from bayes_vi_amd.inference.mcmc.device_hmc import BatchedDeviceNoUTurnSamplerKernel
from bayes_vi_amd.utils.datasets import TensorDataset

n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
c = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d, k = 32, 16
m = d * (d + 1) // 2
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(5)
mu = torch.randint(-20, 21, (k, d), generator=g, device=dev).float()
comp = torch.randint(0, k, (n,), generator=g, device=dev)
Y = mu[comp] + torch.randn(n, d, generator=g, device=dev)
priors = collections.OrderedDict(weights=tfd.Dirichlet(torch.ones(k)), locs=tfd.Normal(torch.zeros(k, d), 30.),
                                 scale_tril=tfd.Normal(torch.zeros(k, d, d), 5.))
model = Model(priors, lambda weights, locs, scale_tril: tfd.MixtureSameFamily(
    mixture_distribution=tfd.Categorical(probs=weights),
    components_distribution=tfd.MultivariateNormalTriL(loc=locs, scale_tril=scale_tril)),
    constraining_bijectors=[tfb.SoftmaxCentered(), tfb.Identity(), tfb.FillScaleTriL(d, 1e-3, batch=k)])
post = MCMC(model, TensorDataset(None, Y)).posterior
sp_inv_1 = float(np.log(np.expm1(1.0 - 1e-3)))
L0 = torch.zeros(k, d, d); L0[:, range(d), range(d)] = 1.0
z0 = torch.cat([torch.zeros(k - 1), mu.cpu().reshape(-1),
                tfb.FillScaleTriL(d, 1e-3, batch=k).inverse(L0.reshape(-1))]).to(dev)
z0 = z0[None].repeat(c, 1) + 0.3 / np.sqrt(n / k) * torch.randn(c, z0.numel(), device=dev, generator=g)
eps = 0.1 / np.sqrt(n / k)
T = 4
blocks = post.device_blocks()
assert blocks is not None
variants = [('host tree, torch assembly', False, False), ('host tree, device assembly', True, False),
            ('device tree, device assembly', True, True)]
for name, dev_asm, dev_tree in variants:
    post._blocks = blocks if dev_asm else None
    kern = (BatchedDeviceNoUTurnSamplerKernel(post, eps, 4, seed=1) if dev_tree
            else samplers.NoUTurnSamplerKernel(post.unconstrained_log_prob_and_grad, eps, 4))
    z = z0.clone()
    st = kern.bootstrap(z)
    gen = torch.Generator(device=dev).manual_seed(1)
    step = (lambda z, st: kern.one_step(z, st)) if dev_tree else (lambda z, st: kern.one_step(z, st, gen))
    z, st, tr = step(z, st)
    torch.cuda.synchronize(); t0 = time.perf_counter(); leaps = 0
    for _ in range(T):
        z, st, tr = step(z, st)
        leaps += int(tr['leapfrogs_taken'].max())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('%-30s N=%d C=%d: %.1f ms/transition, %.1f lock-step leapfrogs, %.2f ms/leapfrog, accept %.2f, lp %.4g'
          % (name, n, c, 1e3 * dt / T, leaps / T, 1e3 * dt / max(leaps, 1),
             float(torch.exp(tr['log_accept_ratio']).mean()), float(st['target_log_prob'].mean())), flush=True)
