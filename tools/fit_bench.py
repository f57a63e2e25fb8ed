"""Wall time of MCMC.fit end to end on a mid-size regression with a handful of chains (the usual way the reference is
used): HMC and NUTS with dual-averaging step-size adaptation.  Usage: fit_bench.py [N] [D] [chains]"""
import collections, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bayesian-inference_amd'))
import bayes_vi_amd.distributions as tfd
from bayes_vi_amd import symbolic
from bayes_vi_amd.model import Model
from bayes_vi_amd.inference.mcmc import MCMC
from bayes_vi_amd.inference.mcmc.transition_kernels import HamiltonianMonteCarlo, NoUTurnSampler
from bayes_vi_amd.inference.mcmc.stepsize_adaptation_kernels import DualAveragingStepSizeAdaptation
from bayes_vi_amd.utils.datasets import TensorDataset

n, d, c = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (1_000_000, 32, 4)
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn(n, d, device=dev, generator=g)
w = torch.randn(d, device=dev, generator=g)
y = X @ w + 0.3 + torch.randn(n, device=dev, generator=g)
priors = collections.OrderedDict(b=tfd.Normal(0., 10.), w=tfd.Normal(torch.zeros(d), 10.), scale=tfd.HalfNormal(10.))
model = Model(priors, lambda b, w, scale, features: tfd.Normal(loc=b + symbolic.matvec(features, w), scale=scale))
ds = TensorDataset(X, y)
init = [torch.full((c,), 0.3), w.cpu().repeat(c, 1), torch.ones(c)]
eps0 = 0.3 / np.sqrt(n)
for name, kern in (('HMC(10) + dual averaging', HamiltonianMonteCarlo(eps0, 10, stepsize_adaptation_kernel=DualAveragingStepSizeAdaptation(200))),
                   ('NUTS + dual averaging', NoUTurnSampler(eps0, max_tree_depth=6, stepsize_adaptation_kernel=DualAveragingStepSizeAdaptation(200)))):
    mc = MCMC(model=model, dataset=ds, transition_kernel=kern)
    mc.fit(initial_state=init, num_samples=5, num_burnin_steps=5, progress_bar=False, seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = mc.fit(initial_state=init, num_samples=1000, num_burnin_steps=300, progress_bar=False, seed=2)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ws = res.samples[1].reshape(-1, d)
    extra = ''
    if 'leapfrogs_taken' in res.trace:
        extra = ', %.1f leapfrogs/transition' % float(res.trace['leapfrogs_taken'].float().mean())
    print('%s: N=%d D=%d, %d chains x 1300 transitions in %.2f s (%.2f ms/transition%s); accept %.2f; |mean w - w| max %.2e (posterior sd %.1e)'
          % (name, n, d, c, dt, 1e3 * dt / 1300, extra, float(res.accept_ratios.mean()), float((ws.mean(0).cpu() - w.cpu()).abs().max()),
             1 / np.sqrt(n)), flush=True)
