"""GPU: the single multivariate Gaussian with a Wishart prior on the Cholesky factor of its precision
(exploratory_experiments/multivariate-gaussian-test.ipynb cell 15): the C-ABI entry point bvi_mvn_tril_logp_grad
against the float64 oracle, the assembled posterior on the device path and on the torch path, and the model through
MCMC.fit / VI.fit.  The reference keeps neither the notebook's data nor a table for this model: parity unpinned; the
oracle is pinned by scipy.stats and finite differences (tests/test_host_logic.py)."""
import collections

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import closed_form as cf
from parity_util import assert_grad_close, assert_logp_close

pytestmark = pytest.mark.gpu


def dev(a, device):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)


def _theta(rng, c, d, mode):
    loc = rng.normal(1.0, 0.3, (c, d))
    M = np.tril(rng.normal(size=(c, d, d)) * (0.15 if mode else 0.3))
    for i in range(d):
        M[:, i, i] = rng.uniform(0.6, 1.4, c)
    return np.concatenate([loc, M.reshape(c, -1)], -1).astype(np.float32)


@pytest.mark.parametrize('d,c,n', [(2, 3, 100), (3, 1, 17), (5, 4, 1000), (8, 2, 4099), (16, 3, 700), (32, 2, 3000)])
@pytest.mark.parametrize('mode', [0, 1])
def test_mvn_tril_entry_point_against_oracle(gpu_device, d, c, n, mode):
    from bayes_vi_amd import _native as nat
    ws = nat.Workspace(torch.device(gpu_device))
    rng = np.random.default_rng(100 * d + 10 * c + mode)
    th = _theta(rng, c, d, mode)
    A = np.tril(rng.normal(size=(d, d)) * 0.4) + np.eye(d)
    Y = (1.0 + rng.normal(size=(n, d)) @ A.T).astype(np.float32)
    ref_lp, ref_g = (cf.mvn_chol_precision if mode else cf.mvn_scale_tril)(Y, th)
    lp, g = nat.mvn_tril_logp_grad(dev(Y, gpu_device), dev(th, gpu_device), mode, ws)
    assert_logp_close(lp.cpu().numpy(), ref_lp)
    # 1e-5 of the chain's largest gradient entry; every row contributes O(1)-sized terms (times the map's own
    # derivative for the Cholesky-precision form: entries of Sigma (G + G^T) Sigma P are sums of d^2 such products)
    assert_grad_close(g.cpu().numpy(), ref_g, term_scale=np.full_like(ref_g, 3.0 * n))
    gm = g.cpu().numpy()[:, d:].reshape(c, d, d)
    assert np.all(np.triu(gm, 1) == 0)                    # the gradient of the matrix parameter is lower triangular
    lp2, g2 = nat.mvn_tril_logp_grad(dev(Y, gpu_device), dev(th, gpu_device), mode, ws, want_grad=False)
    assert g2 is None
    np.testing.assert_array_equal(lp2.cpu().numpy(), lp.cpu().numpy())
    # entries above the diagonal of the matrix parameter are not read
    th2 = th.copy().reshape(c, -1)
    M = th2[:, d:].reshape(c, d, d); M += np.triu(np.full((d, d), 7.0), 1).astype(np.float32)
    lp3, g3 = nat.mvn_tril_logp_grad(dev(Y, gpu_device), dev(th2, gpu_device), mode, ws)
    assert torch.equal(lp3, lp) and torch.equal(g3, g)


def test_mvn_tril_entry_point_rejects_bad_arguments(gpu_device):
    from bayes_vi_amd import _native as nat
    lib = nat.lib()
    Y = torch.zeros(4, 2, device=gpu_device); th = torch.ones(1, 6, device=gpu_device)
    lp = torch.zeros(1, device=gpu_device)
    buf = torch.zeros(1 << 20, dtype=torch.uint8, device=gpu_device)
    args = lambda d, mode, nbytes: (nat._ptr(Y), 4, d, mode, nat._ptr(th), 1, nat._ptr(lp), None, None, nat._ptr(buf), nbytes, nat._stream())
    assert lib.bvi_mvn_tril_logp_grad(*args(2, 2, buf.numel())) == -1          # unknown parameterisation
    assert lib.bvi_mvn_tril_logp_grad(*args(33, 0, buf.numel())) == -2         # beyond 32 dimensions
    assert lib.bvi_mvn_tril_logp_grad(*args(2, 0, 16)) == -3                   # workspace too small
    assert float(lp.abs().sum()) == 0.0


def _notebook_model(dims, wishart_scale=None):
    import bayes_vi_amd.bijectors as tfb
    import bayes_vi_amd.distributions as tfd
    import bayes_vi_amd.linalg as tfl
    from bayes_vi_amd.model import Model
    priors = collections.OrderedDict(
        loc=tfd.MultivariateNormalDiag(loc=[0.] * dims, scale_diag=[10.] * dims),
        precision=tfd.WishartTriL(df=5, scale_tril=10.0 * torch.eye(dims) if wishart_scale is None else wishart_scale,
                                  input_output_cholesky=True))

    def likelihood(loc, precision):
        return tfd.MultivariateNormalTriL(loc=loc, scale_tril=tfl.cholesky_solve(precision, tfl.eye(dims)))

    return Model(priors=priors, likelihood=likelihood,
                 constraining_bijectors=[tfb.Identity(), tfb.FillScaleTriL(diag_bijector=tfb.Softplus(), diag_shift=1e-3)])


def _notebook_data(n, seed=0, S=((3.0, 0.0), (7.0, 4.0))):
    rng = np.random.default_rng(seed)
    return (np.array([1.0, 7.0]) + rng.normal(size=(n, 2)) @ np.array(S).T).astype(np.float32)


def _dataset(Y):
    from bayes_vi_amd.utils.datasets import make_dataset_from_df
    names = ['y%d' % (i + 1) for i in range(Y.shape[1])]
    return make_dataset_from_df(pd.DataFrame({k: Y[:, i] for i, k in enumerate(names)}), target_names=names,
                                format_features_as='dict')


def test_mvn_wishart_posterior_matches_oracle_on_device_and_torch_paths(gpu_device):
    from bayes_vi_amd.inference.mcmc import MCMC
    rng = np.random.default_rng(4)
    for dims, n in ((2, 1000), (4, 300)):
        A = np.tril(rng.normal(size=(dims, dims)) * 0.5) + np.eye(dims) * 2
        Y = (rng.normal(size=(n, dims)) @ A.T + 1.0).astype(np.float32) if dims != 2 else _notebook_data(n)
        z = rng.normal(0, 0.3, (5, dims + dims * (dims + 1) // 2)).astype(np.float32)
        zt = torch.tensor(z, device=gpu_device)
        spec = cf.mvn_chol_precision_model(Y)
        rlp, rg = spec.log_post_unconstrained(z)
        model = _notebook_model(dims)
        post = MCMC(model, _dataset(Y)).posterior
        assert post.device_blocks() is not None                      # priors and bijectors assembled on the device
        lp, g = post.unconstrained_log_prob_and_grad(zt)
        assert_logp_close(lp.cpu().numpy(), rlp); assert_grad_close(g.cpu().numpy(), rg)
        # the same model with priors and chain rule composed from the torch objects
        post._blocks = None
        lp2, g2 = post.unconstrained_log_prob_and_grad(zt)
        assert_logp_close(lp2.cpu().numpy(), rlp); assert_grad_close(g2.cpu().numpy(), rg)
        # a Wishart scale the device prior does not cover goes the torch way by itself
        ws = torch.tensor(np.tril(rng.normal(size=(dims, dims))) + 8.0 * np.eye(dims), dtype=torch.float32)
        post3 = MCMC(_notebook_model(dims, ws), _dataset(Y)).posterior
        assert post3.device_blocks() is None
        spec3 = cf.mvn_chol_precision_model(Y)
        spec3.blocks[1]['prior'] = ('wishart_tril', 5.0, ws.double().numpy())
        rlp3, rg3 = spec3.log_post_unconstrained(z)
        lp3, g3 = post3.unconstrained_log_prob_and_grad(zt)
        assert_logp_close(lp3.cpu().numpy(), rlp3); assert_grad_close(g3.cpu().numpy(), rg3)


def test_mvn_wishart_model_through_mcmc_and_vi(gpu_device):
    """The notebook's run in miniature: 1000 points from N([1, 7], S S^T).  NUTS on the device (tree bookkeeping and
    posterior assembly on the GPU, the data pass through bvi_mvn_tril_logp_grad): the posterior mean of loc sits at
    the sample mean and that of the covariance the model evaluates, tril(Sigma) tril(Sigma)^T with Sigma =
    (P P^T)^-1, at the sample covariance, within Monte-Carlo error and the prior's pull.  Mean-field ADVI through
    VI.fit lowers its loss; the posterior predictive draws have the data's mean.
    The scale the data are drawn with is [[3, 0], [1, 4]], not the notebook's [[3, 0], [7, 4]]: the model's scale is
    the lower triangle [[a, 0], [b, c]] of a positive definite [[a, b], [b, c]], which needs a c > b^2, and 3 * 4 <
    7^2 -- on the notebook's own data the posterior sits on that boundary (its printed posterior mean of P gives
    a c = 22.1 against b^2 = 21.9), away from the sample covariance."""
    from bayes_vi_amd.inference.mcmc import MCMC
    from bayes_vi_amd.inference.mcmc.stepsize_adaptation_kernels import DualAveragingStepSizeAdaptation
    from bayes_vi_amd.inference.mcmc.transition_kernels import NoUTurnSampler
    from bayes_vi_amd.inference.vi import VI
    from bayes_vi_amd.inference.vi.surrogate_posteriors import ADVI
    from bayes_vi_amd.inference.vi.vi import Adam
    Y = _notebook_data(1000, seed=3, S=((3.0, 0.0), (1.0, 4.0)))
    model, ds = _notebook_model(2), _dataset(Y)
    P0 = np.linalg.cholesky(np.linalg.inv(np.cov(Y.T)))
    # a chain started at tril(Sigma) = the sample covariance's Cholesky factor would need the map inverted; start
    # near the precision factor of the sample covariance and let the burn-in find the model's own mode
    init = [torch.tensor(np.tile(Y.mean(0), (4, 1)), dtype=torch.float32),
            torch.tensor(np.tile(P0, (4, 1, 1)), dtype=torch.float32)]
    kernel = NoUTurnSampler(step_size=0.01, max_tree_depth=6,
                            stepsize_adaptation_kernel=DualAveragingStepSizeAdaptation(num_adaptation_steps=600))
    mcmc = MCMC(model=model, dataset=ds, transition_kernel=kernel)
    res = mcmc.fit(initial_state=init, num_samples=800, num_burnin_steps=800, progress_bar=False, seed=11)
    loc, P = (s.cpu().numpy() for s in res.samples)
    assert loc.shape == (800, 4, 2) and P.shape == (800, 4, 2, 2)
    assert np.all(P[..., 0, 1] == 0) and np.all(P[..., 0, 0] > 0) and np.all(P[..., 1, 1] > 0)
    assert float(res.accept_ratios.min()) > 0.4
    se = Y.std(0) / np.sqrt(len(Y))
    np.testing.assert_allclose(loc.reshape(-1, 2).mean(0), Y.mean(0), atol=4 * se.max() + 0.05)
    Sigma = np.linalg.inv(P @ np.swapaxes(P, -1, -2))
    S = np.tril(Sigma)
    cov = (S @ np.swapaxes(S, -1, -2)).reshape(-1, 2, 2).mean(0)
    np.testing.assert_allclose(cov, np.cov(Y.T), rtol=0.12, atol=1.0)
    # mean-field ADVI on the same model
    torch.manual_seed(0)
    vi = VI(model=model, dataset=ds, surrogate_posterior=ADVI(model=model, mean_field=True))
    approx, losses = vi.fit(optimizer=Adam(learning_rate=2e-2), num_steps=600, sample_size=10,
                            num_samples_to_approx_marginals=500, progress_bar=False)
    assert np.isfinite(losses).all() and losses[-50:].mean() < losses[:50].mean()
    assert approx['precision'].mean().shape == (2, 2)
    # posterior predictive (cell 25): parameter draws and replicated data
    pred = model.get_posterior_predictive_distribution(posterior_samples=[s.reshape((-1,) + tuple(s.shape[2:])).cpu()
                                                                          for s in res.samples])
    draws = pred.sample((2000,), generator=torch.Generator().manual_seed(2))
    assert draws['y'].shape[0] == 2000 and draws['y'].shape[-1] == 2
    np.testing.assert_allclose(draws['y'].reshape(-1, 2).mean(0).numpy(), Y.mean(0), atol=0.6)
