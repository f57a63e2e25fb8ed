"""HMC transitions per second on the 20-point Gaussian of BASELINE.json configs[0]: whole run in one launch
(bvi_hmc_fused) against the launch-per-leapfrog-step device path."""
import collections, os, sys, time
import numpy as np, pandas as pd, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
import bayes_vi_amd.distributions as tfd
from bayes_vi_amd.model import Model
from bayes_vi_amd.inference.mcmc import MCMC
from bayes_vi_amd.utils.datasets import make_dataset_from_df
from bayes_vi_amd.inference.mcmc.device_hmc import (DeviceHamiltonianMonteCarloKernel, FusedHamiltonianMonteCarloKernel,
                                                    fused_family)

rng = np.random.default_rng(0)
for n in (20, 2000, 50000):
    y = rng.normal(7, 3, n).astype(np.float32)
    priors = collections.OrderedDict(loc=tfd.Normal(loc=0., scale=10.), scale=tfd.HalfNormal(scale=10.))
    model = Model(priors=priors, likelihood=lambda loc, scale: tfd.Normal(loc=loc, scale=scale))
    ds = make_dataset_from_df(pd.DataFrame(dict(y=y)), target_names=['y'], format_features_as='dict')
    post = MCMC(model, ds).posterior
    for c in (2, 64, 1024):
        z = torch.tensor(rng.uniform(-.5, .5, (c, 2)) + np.array([7.0, 1.0]), dtype=torch.float32, device='cuda:0')
        eps = 0.5 / np.sqrt(n)
        fused = FusedHamiltonianMonteCarloKernel(post, fused_family(post), step_size=eps, num_leapfrog_steps=5)
        st = fused.bootstrap(z)
        T = 4000 if n <= 2000 else 400
        fused.many_steps(z, st, 10); torch.cuda.synchronize()
        t0 = time.perf_counter(); zs, zf, stf, tr = fused.many_steps(z, st, T); torch.cuda.synchronize()
        tf = time.perf_counter() - t0
        step = DeviceHamiltonianMonteCarloKernel(post, step_size=eps, num_leapfrog_steps=5)
        ss = step.bootstrap(z); zz = z
        for _ in range(5):
            zz, ss, _ = step.one_step(zz, ss)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            zz, ss, _ = step.one_step(zz, ss)
        torch.cuda.synchronize()
        ts = (time.perf_counter() - t0) / 100
        print('n=%6d chains=%5d: fused %8.1f us/transition (accept %.2f) | per-step launches %8.1f us/transition | x%.0f'
              % (n, c, 1e6 * tf / T, float(tr['is_accepted'].float().mean()), 1e6 * ts, ts / (tf / T)), flush=True)
