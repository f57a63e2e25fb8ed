"""Wall time of the reference's small VI runs (README.md:265-285, 626-643; vi_summary_statistics/final_summary.txt):
mean-field ADVI and Hamiltonian flows on the
univariate Gaussian (N = 20, S = 10, 1000 steps; N = 100, S = 50, 10 000 steps)."""
import collections, os, sys, time
import numpy as np, pandas as pd, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
import bayes_vi_amd.distributions as tfd
from bayes_vi_amd.model import Model
from bayes_vi_amd.inference.vi import VI
from bayes_vi_amd.inference.vi.surrogate_posteriors import ADVI, NormalizingFlow
from bayes_vi_amd.inference.vi.flow_bijectors import HamiltonianFlow
from bayes_vi_amd.inference.vi.vi import Adam
from bayes_vi_amd.utils.datasets import make_dataset_from_df

rng = np.random.default_rng(0)
priors = collections.OrderedDict(loc=tfd.Normal(loc=0., scale=10.), scale=tfd.HalfNormal(scale=10.))
model = Model(priors=priors, likelihood=lambda loc, scale: tfd.Normal(loc=loc, scale=scale))
for n, S, steps, ref in ((20, 10, 1000, 'ADVI 2 s, HNF(2 leapfrogs) 7 s'), (100, 50, 10000, 'ADVI 10.1 s, HNF(1/2/3) 57.9/93.5/132.2 s')):
    y = rng.normal(7, 3, n).astype(np.float32)
    ds = make_dataset_from_df(pd.DataFrame(dict(y=y)), target_names=['y'], format_features_as='dict')
    nd = model.flat_unconstrained_param_event_ndims
    cases = [('mean-field ADVI', lambda: ADVI(model, mean_field=True))]
    for L in ((2,) if n == 20 else (1, 2, 3)):
        cases.append(('HNF %d leapfrog' % L, lambda L=L: NormalizingFlow(
            model=model, flow_bijector=HamiltonianFlow(nd, step_sizes=torch.nn.Parameter(torch.tensor(0.1)),
                                                      num_integration_steps=L), extra_ndims=nd)))
    for name, make in cases:
        vi = VI(model, ds, make())
        vi.fit(optimizer=Adam(1e-2), num_steps=20, sample_size=S, num_samples_to_approx_marginals=10, progress_bar=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, losses = vi.fit(optimizer=Adam(1e-2), num_steps=steps, sample_size=S, num_samples_to_approx_marginals=1000,
                           progress_bar=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print('N=%d S=%d %s: %d steps in %.2f s (%.3f ms/step); loss %.3f -> %.3f   [reference: %s]'
              % (n, S, name, steps, dt, 1e3 * dt / steps, losses[:20].mean(), losses[-100:].mean(), ref), flush=True)
