"""BASELINE.json configs[0]: univariate Gaussian, 20 points, NUTS + dual averaging, 2 chains x 4000 samples
(README.md:146-162): wall time of MCMC.fit, and of HMC on the same posterior."""
import collections, os, sys, time
import numpy as np, pandas as pd, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
import bayes_vi_amd.distributions as tfd
from bayes_vi_amd.model import Model
from bayes_vi_amd.inference.mcmc import MCMC
from bayes_vi_amd.inference.mcmc.transition_kernels import NoUTurnSampler, HamiltonianMonteCarlo
from bayes_vi_amd.inference.mcmc.stepsize_adaptation_kernels import DualAveragingStepSizeAdaptation
from bayes_vi_amd.utils.datasets import make_dataset_from_df

rng = np.random.default_rng(1)
y = rng.normal(7, 3, 20).astype(np.float32)
priors = collections.OrderedDict(loc=tfd.Normal(loc=0., scale=10.), scale=tfd.HalfNormal(scale=10.))
model = Model(priors=priors, likelihood=lambda loc, scale: tfd.Normal(loc=loc, scale=scale))
ds = make_dataset_from_df(pd.DataFrame(dict(y=y)), target_names=['y'], format_features_as='dict')
chains = int(os.environ.get('CHAINS', 2))
for name, kern in (('NUTS + dual averaging', NoUTurnSampler(stepsize_adaptation_kernel=DualAveragingStepSizeAdaptation(800))),
                   ('HMC(8) + dual averaging', HamiltonianMonteCarlo(0.1, 8, stepsize_adaptation_kernel=DualAveragingStepSizeAdaptation(800)))):
    mc = MCMC(model=model, dataset=ds, transition_kernel=kern)
    mc.fit(num_chains=chains, num_samples=20, num_burnin_steps=20, progress_bar=False, seed=1)   # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = mc.fit(num_chains=chains, num_samples=4000, num_burnin_steps=1000, progress_bar=False, seed=2)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    loc, sc = (s.cpu().numpy() for s in res.samples)
    extra = ''
    if 'leapfrogs_taken' in res.trace:
        extra = ' leapfrogs/transition %.1f' % float(res.trace['leapfrogs_taken'].float().mean())
    print('%s: %d chains x 5000 transitions in %.2f s (%.2f ms/transition)%s; loc %.3f +- %.3f scale %.3f +- %.3f'
          % (name, chains, dt, 1e3 * dt / 5000, extra, loc.mean(), loc.std(), sc.mean(), sc.std()), flush=True)

# the reference's experiment runs: 5 chains x 11 000 NUTS steps, N = 100
# (univariate_gaussian_experiment and linear_regression_experiment, acceptance_and_runtime_summary.txt:6-15)
rng = np.random.default_rng(3)
y100 = rng.normal(7, 3, 100).astype(np.float32)
ds = make_dataset_from_df(pd.DataFrame(dict(y=y100)), target_names=['y'], format_features_as='dict')
kern = NoUTurnSampler(stepsize_adaptation_kernel=DualAveragingStepSizeAdaptation(800))
mc = MCMC(model=model, dataset=ds, transition_kernel=kern)
mc.fit(num_chains=5, num_samples=20, num_burnin_steps=20, progress_bar=False, seed=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
res = mc.fit(num_chains=5, num_samples=10000, num_burnin_steps=1000, progress_bar=False, seed=2)
torch.cuda.synchronize()
print('univariate Gaussian N=100, NUTS, 5 chains x 11000 steps: %.2f s (reference: 34.7-65.7 s)' % (time.perf_counter() - t0), flush=True)
x = rng.normal(0, 10, 100).astype(np.float32); yy = (7 + 3 * x + rng.normal(0, 10, 100)).astype(np.float32)
priors = collections.OrderedDict(beta_0=tfd.Normal(0., 10.), beta_1=tfd.Normal(0., 10.), scale=tfd.HalfNormal(10.))
model2 = Model(priors, lambda beta_0, beta_1, scale, features: tfd.Normal(loc=beta_0 + beta_1 * features['x'], scale=scale))
ds2 = make_dataset_from_df(pd.DataFrame(dict(x=x, y=yy)), target_names=['y'], feature_names=['x'])
mc = MCMC(model=model2, dataset=ds2, transition_kernel=NoUTurnSampler(stepsize_adaptation_kernel=DualAveragingStepSizeAdaptation(800)))
mc.fit(num_chains=5, num_samples=20, num_burnin_steps=20, progress_bar=False, seed=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
res = mc.fit(num_chains=5, num_samples=10000, num_burnin_steps=1000, progress_bar=False, seed=2)
torch.cuda.synchronize()
b1 = res.samples[1].cpu().numpy()
print('1-feature regression N=100, NUTS, 5 chains x 11000 steps: %.2f s (reference: 184-284 s); slope %.3f +- %.3f, leapfrogs/transition %.1f'
      % (time.perf_counter() - t0, b1.mean(), b1.std(), float(res.trace['leapfrogs_taken'].float().mean())), flush=True)
