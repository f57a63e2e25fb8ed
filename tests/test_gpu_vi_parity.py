"""GPU: the variational-inference rows of the hot path against the fp64 oracle (oracle/elbo.py), through VI's own
code: the target closure on flat constrained samples and its autograd gradient, the lift correction of augmented
flows, the ADVI sampler and its log density, and whole optimisation steps (loss, per-variable clipped gradients,
Adam update) on injected noise.  Reference: bayes_vi/inference/vi/vi.py:46-75, 109-116 and
bayes_vi/inference/vi/surrogate_posteriors.py:72-94, 113-134, 153-212.

Tolerances: parity_util's (log density 1e-5 relative; gradients 1e-5 of the variable's largest entry).  Updated
parameters are compared on the size of the update, lr: an Adam step is lr x a ratio of gradient moments, so a
relative gradient error e moves it by ~lr e.
"""
import collections
import math

import numpy as np
import pytest
import torch

from oracle import closed_form as cf
from oracle import elbo
from parity_util import RTOL, assert_grad_close, assert_logp_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def bvi(gpu_device):
    import bayes_vi_amd.distributions as tfd
    from bayes_vi_amd import symbolic
    from bayes_vi_amd.inference.vi import VI
    from bayes_vi_amd.inference.vi.flow_bijectors import Chain, HamiltonianFlow, make_scale_fn, make_shift_fn
    from bayes_vi_amd.inference.vi.surrogate_posteriors import ADVI, NormalizingFlow
    from bayes_vi_amd.inference.vi.vi import Adam
    from bayes_vi_amd.model import Model
    from bayes_vi_amd.utils.datasets import TensorDataset
    ns = dict(locals()); ns.pop('gpu_device')
    return collections.namedtuple('bvi', ns.keys())(**ns)


def regression(bvi, dev, n, d, seed):
    """Regression model with a D-vector of slopes (README.md:747-756 generalised) and the oracle's description of it;
    flat parameter order (b, w[D], scale) on both sides."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, d)).astype(np.float32)
    w = rng.normal(size=d)
    y = (0.5 + X @ w + rng.normal(size=n)).astype(np.float32)
    priors = collections.OrderedDict(b=bvi.tfd.Normal(0., 10.), w=bvi.tfd.Normal(torch.zeros(d), 10.),
                                     scale=bvi.tfd.HalfNormal(10.))
    model = bvi.Model(priors, lambda b, w, scale, features: bvi.tfd.Normal(
        loc=b + bvi.symbolic.matvec(features, w), scale=scale))
    ds = bvi.TensorDataset(torch.tensor(X, device=dev), torch.tensor(y, device=dev))
    return model, ds, cf.linreg_model(X, y), (X, y)


def inject(surrogate, noise, dev):
    """Make the surrogate's reparameterised sampler transform `noise` (a list of [S, dim] arrays, one per call)."""
    queue = [torch.tensor(np.asarray(e, np.float32), device=dev) for e in noise]

    def base_noise(n, dim):
        e = queue.pop(0)
        assert tuple(e.shape) == (n, dim)
        return e
    surrogate.base_noise = base_noise


def advi_params(sur):
    p = {'loc': sur.loc.detach().cpu().double().numpy()}
    if sur.mean_field:
        p['raw_scale'] = sur.raw_scale.detach().cpu().double().numpy()
    else:
        p['raw_tril'] = sur.raw_tril.detach().cpu().double().numpy()
    return p


def f32_noise(rng, shape):
    return rng.normal(size=shape).astype(np.float32)


def test_vi_closure_value_and_gradient_on_flat_samples(bvi, gpu_device):
    """vi.py:46-52 through surrogate_posteriors.py:89-92: target_log_prob(sample[S, P]) -> [S], differentiated by
    autograd as the GradientTape of vi.py:66-69 does."""
    for d, s in ((3, 8), (64, 256)):
        model, ds, spec, _ = regression(bvi, gpu_device, 20_000, d, seed=d)
        vi = bvi.VI(model, ds, bvi.ADVI(model, mean_field=True))
        rng = np.random.default_rng(100 + d)
        theta = np.concatenate([rng.normal(0.5, 0.3, (s, 1)), rng.normal(0, 1, (s, d)), rng.uniform(0.5, 3, (s, 1))],
                               -1).astype(np.float32)
        t = torch.tensor(theta, device=gpu_device, requires_grad=True)
        lp = vi.target_log_prob(t)
        assert tuple(lp.shape) == (s,)
        wgt = torch.linspace(0.5, 1.5, s, device=gpu_device)          # an upstream gradient that differs per sample
        (lp * wgt).sum().backward()
        ref_lp, ref_g = spec.log_post_constrained(theta)
        assert_logp_close(lp.detach().cpu().numpy(), ref_lp)
        assert_grad_close(t.grad.cpu().numpy(), ref_g * wgt.cpu().double().numpy()[:, None])


def test_augmented_closure_adds_the_lift_log_density(bvi, gpu_device):
    """surrogate_posteriors.py:79-88: with extra dimensions the target is log pi(q) + lift(unconstrain(q)).log_prob(a);
    default lift N(0, I), and an MLP-parameterised one as in linear-regression-example.ipynb cell 13."""
    d, s = 3, 16
    model, ds, spec, _ = regression(bvi, gpu_device, 5_000, d, seed=11)
    pu = model.flat_unconstrained_param_event_ndims
    rng = np.random.default_rng(12)
    theta = np.concatenate([rng.normal(0.5, 0.3, (s, 1)), rng.normal(0, 1, (s, d)), rng.uniform(0.5, 3, (s, 1))], -1)
    a = rng.normal(size=(s, pu))
    sample = torch.tensor(np.concatenate([theta, a], -1), dtype=torch.float32, device=gpu_device)
    ref_lp, _ = spec.log_post_constrained(sample[:, :pu].cpu().numpy())
    a32 = sample[:, pu:].cpu().double().numpy()
    # default lift
    flow = bvi.HamiltonianFlow(pu, num_integration_steps=1, hidden_layers=[8])
    sur = bvi.NormalizingFlow(model, flow_bijector=flow, extra_ndims=pu)
    vi = bvi.VI(model, ds, sur)
    got = vi.target_log_prob(sample).detach().cpu().numpy()
    assert_logp_close(got, ref_lp + (-0.5 * a32 * a32 - 0.5 * elbo.LN2PI).sum(-1))
    # MLP lift: loc = shift_fn(z), scale_diag = scale_fn(z) with z the unconstrained parameters
    torch.manual_seed(0)
    shift_fn, scale_fn = bvi.make_shift_fn(pu, [8]), bvi.make_scale_fn(pu, [8])
    lift = lambda q: bvi.tfd.MultivariateNormalDiag(loc=shift_fn(q), scale_diag=scale_fn(q))
    sur = bvi.NormalizingFlow(model, flow_bijector=flow, extra_ndims=pu, posterior_lift_distribution=lift)
    vi = bvi.VI(model, ds, sur)
    assert sum(p.numel() for p in sur.lift_networks.parameters()) > 0      # trained with the surrogate (vi.py:69)
    got = vi.target_log_prob(sample).detach().cpu().numpy()
    z = spec.unconstrain(sample[:, :pu].cpu().double().numpy())
    layers = lambda net: [(torch.tensor(m.weight.detach().cpu().double().numpy().T),
                           torch.tensor(m.bias.detach().cpu().double().numpy()))
                          for m in net if isinstance(m, torch.nn.Linear)]
    zt = torch.tensor(z)
    shift = elbo.t_mlp(zt, layers(shift_fn), 'relu6', None).numpy()
    scale = elbo.t_mlp(zt, layers(scale_fn), 'relu6', 'softplus').numpy()
    lift_lp = (-0.5 * ((a32 - shift) / scale) ** 2 - 0.5 * elbo.LN2PI - np.log(scale)).sum(-1)
    assert_logp_close(got, ref_lp + lift_lp)


@pytest.mark.parametrize('mean_field', [True, False])
def test_advi_sampler_and_log_density(bvi, gpu_device, mean_field):
    """surrogate_posteriors.py:113-134: constrained draws of the surrogate and log q including the inverse
    log-det-Jacobian of the constraining bijector."""
    d, s = 5, 64
    model, ds, spec, _ = regression(bvi, gpu_device, 1_000, d, seed=21)
    torch.manual_seed(3)
    sur = bvi.ADVI(model, mean_field=mean_field).to(gpu_device)
    with torch.no_grad():        # away from the initial state, so that every term matters
        sur.loc.add_(0.3)
        if mean_field:
            sur.raw_scale.add_(torch.linspace(-1, 1, sur.raw_scale.numel(), device=gpu_device))
        else:
            sur.raw_tril.add_(0.1 * torch.randn_like(sur.raw_tril))
    eps = f32_noise(np.random.default_rng(22), (s, spec.pu))
    inject(sur, [eps], gpu_device)
    q, logq = sur.sample_and_log_prob(s)
    _, theta, ref_logq = elbo.advi_sample_and_log_prob(spec, advi_params(sur), eps, mean_field)
    np.testing.assert_allclose(q.detach().cpu().numpy(), theta, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(logq.detach().cpu().numpy(), ref_logq, rtol=RTOL, atol=1e-5)
    # initial state of the reference: loc ~ N(0, 1), unit scale
    fresh = bvi.ADVI(model, mean_field=mean_field)
    L = torch.diag(torch.nn.functional.softplus(fresh.raw_scale)) if mean_field else fresh.scale_tril()
    np.testing.assert_allclose(L.detach().numpy(), np.eye(spec.pu), atol=1e-6)


def run_steps(vi, sur, opt_desc, s, n_steps):
    params = [p for p in sur.parameters() if p.requires_grad]
    opt = opt_desc.bind(params)
    out = []
    for _ in range(n_steps):
        loss = vi._step(opt, params, s)
        out.append((float(loss), {n: p.grad.detach().cpu().double().numpy().copy() for n, p in sur.named_parameters()},
                    {n: p.detach().cpu().double().numpy().copy() for n, p in sur.named_parameters()}))
    return out


@pytest.mark.parametrize('d,s,mean_field', [(3, 8, True), (64, 256, True), (64, 8, True), (3, 256, True),
                                            (3, 8, False), (64, 256, False), (3, 256, False)])
def test_advi_optimisation_steps_match_oracle(bvi, gpu_device, d, s, mean_field):
    """Three consecutive `optimizer_step`s of vi.py:64-73 on the same noise as the oracle: the Monte-Carlo loss
    (vi.py:109-116), every variable's gradient after clip_by_norm(., 1) (vi.py:70) and the variables after Adam."""
    model, ds, spec, _ = regression(bvi, gpu_device, 20_000, d, seed=30 + d)
    torch.manual_seed(5)
    sur = bvi.ADVI(model, mean_field=mean_field)
    vi = bvi.VI(model, ds, sur)
    sur = vi.surrogate_posterior
    rng = np.random.default_rng(40 + d + s)
    noise = [f32_noise(rng, (s, spec.pu)) for _ in range(3)]
    inject(sur, noise, gpu_device)
    lr = 0.05
    params = advi_params(sur)
    ref_opt = elbo.Adam(params, learning_rate=lr)
    got = run_steps(vi, sur, bvi.Adam(learning_rate=lr), s, 3)
    raw_name = 'raw_scale' if mean_field else 'raw_tril'
    for step in range(3):
        ref_loss, ref_clipped, params = elbo.advi_step(spec, params, noise[step], ref_opt, mean_field)
        loss, grads, new = got[step]
        assert_logp_close(np.array([loss]), np.array([ref_loss]))
        for name in ('loc', raw_name):
            assert_grad_close(grads[name].reshape(1, -1), ref_clipped[name].reshape(1, -1))
            assert np.linalg.norm(grads[name]) <= 1.0 + 1e-6
            np.testing.assert_allclose(new[name], params[name], rtol=0, atol=lr * 20 * RTOL + 1e-6)
        # carry on from the device's own variables, so that steps are compared one by one
        params = {k: new[k] for k in params}


def flow_variables(sur, n_flows, n_lift_layers):
    """The surrogate's trainable tensors under the oracle's names (Dense kernels are [in, out])."""
    v = {}
    flows = list(sur.flow_bijector.bijectors)
    for i, f in enumerate(flows):
        v['flow%d.step' % i] = f.step_sizes
        for e, net in (('U', f.potential_energy_fn), ('T', f.kinetic_energy_fn)):
            for j, m in enumerate([m for m in net if isinstance(m, torch.nn.Linear)]):
                v['flow%d.%s.%d.k' % (i, e, j)] = m.weight
                v['flow%d.%s.%d.b' % (i, e, j)] = m.bias
    if n_lift_layers:
        shift_fn, scale_fn = sur.lift_networks[0], sur.lift_networks[1]
        for name, net in (('shift', shift_fn), ('scale', scale_fn)):
            for j, m in enumerate([m for m in net if isinstance(m, torch.nn.Linear)]):
                v['lift.%s.%d.k' % (name, j)] = m.weight
                v['lift.%s.%d.b' % (name, j)] = m.bias
    return v


def to_oracle(name, t):
    a = t.detach().cpu().double().numpy()
    return a.T.copy() if name.endswith('.k') else a.copy()


@pytest.mark.parametrize('d,s,lift', [(3, 8, True), (64, 256, True), (3, 32, False)])
def test_hamiltonian_flow_optimisation_steps_match_oracle(bvi, gpu_device, d, s, lift):
    """BASELINE configs[4], second half: NormalizingFlow over Chain([HamiltonianFlow] x 2), two leapfrog steps each,
    augmented with Pu extra dimensions and an MLP lift distribution (linear-regression-example.ipynb cell 13), one
    trainable step size per flow.  Two optimisation steps against the fp64 autograd oracle."""
    model, ds, spec, data = regression(bvi, gpu_device, 10_000, d, seed=50 + d)
    pu = model.flat_unconstrained_param_event_ndims
    torch.manual_seed(7)
    hidden = [16, 16]
    flows = [bvi.HamiltonianFlow(pu, step_sizes=torch.nn.Parameter(torch.tensor(0.1)), num_integration_steps=2,
                                 hidden_layers=hidden) for _ in range(2)]
    if lift:
        shift_fn, scale_fn = bvi.make_shift_fn(pu, [16]), bvi.make_scale_fn(pu, [16])
        lift_fn = lambda q, shift_fn=shift_fn, scale_fn=scale_fn: bvi.tfd.MultivariateNormalDiag(
            loc=shift_fn(q), scale_diag=scale_fn(q))
    else:
        lift_fn = None
    sur = bvi.NormalizingFlow(model, flow_bijector=bvi.Chain(flows), extra_ndims=pu,
                              posterior_lift_distribution=lift_fn)
    vi = bvi.VI(model, ds, sur)
    sur = vi.surrogate_posterior
    rng = np.random.default_rng(60 + d)
    noise = [f32_noise(rng, (s, 2 * pu)) for _ in range(2)]
    inject(sur, noise, gpu_device)
    lr = 0.01
    tensors = flow_variables(sur, 2, 2 if lift else 0)
    assert len(tensors) == sum(1 for _ in sur.parameters())
    oracle_sur = elbo.FlowSurrogate(spec.blocks, n_flows=2, n_steps=2, n_energy_layers=3, n_lift_layers=2 if lift else 0)
    post = elbo.t_log_posterior('linreg', data, spec.blocks)
    variables = {k: to_oracle(k, t) for k, t in tensors.items()}
    ref_opt = elbo.Adam(variables, learning_rate=lr)
    params = [p for p in sur.parameters() if p.requires_grad]
    opt = bvi.Adam(learning_rate=lr).bind(params)
    for step in range(2):
        ref_loss, ref_grads = elbo.t_loss_and_grads(lambda leaves: oracle_sur.loss(leaves, post, noise[step]), variables)
        ref_clipped = {k: elbo.clip_by_norm(g) for k, g in ref_grads.items()}
        ref_new = ref_opt.step(variables, ref_clipped)
        loss = float(vi._step(opt, params, s))
        assert_logp_close(np.array([loss]), np.array([ref_loss]))
        for k, t in tensors.items():
            g = to_oracle(k, t.grad)
            scale = max(np.abs(ref_clipped[k]).max(), 1e-3)
            assert np.abs(g - ref_clipped[k]).max() <= 5 * RTOL * scale + 1e-6, (k, np.abs(g - ref_clipped[k]).max(), scale)
            np.testing.assert_allclose(to_oracle(k, t), ref_new[k], rtol=0, atol=lr * 1e-3 + 1e-6, err_msg=k)
        variables = {k: to_oracle(k, t) for k, t in tensors.items()}


def test_non_finite_samples_are_dropped_on_the_device(bvi, gpu_device):
    """vi.py:55-58: a sample whose discrepancy is not finite leaves the mean -- loss and gradients are those of the
    remaining samples."""
    d, s = 3, 8
    model, ds, spec, _ = regression(bvi, gpu_device, 2_000, d, seed=70)
    torch.manual_seed(9)
    vi = bvi.VI(model, ds, bvi.ADVI(model, mean_field=True))
    sur = vi.surrogate_posterior
    eps = f32_noise(np.random.default_rng(71), (s, spec.pu))
    eps[5, -1] = -3000.0          # scale parameter underflows to 0 for this draw: log density is not finite
    inject(sur, [eps], gpu_device)
    loss = vi.loss(s)
    loss.backward()
    with np.errstate(all='ignore'):
        ref_loss, ref_grads, aux = elbo.advi_loss_and_grads(spec, advi_params(sur), eps, True)
    assert aux['finite'].sum() == s - 1
    assert_logp_close(np.array([float(loss.detach())]), np.array([ref_loss]))
    assert_grad_close(sur.loc.grad.cpu().numpy()[None], ref_grads['loc'][None])
    assert_grad_close(sur.raw_scale.grad.cpu().numpy()[None], ref_grads['raw_scale'][None])


@pytest.mark.parametrize('mean_field', [True, False])
def test_sample_with_finite_density_and_non_finite_gradient_is_dropped(bvi, gpu_device, mean_field):
    """A draw whose scale lands at ~1e-13 has a finite float32 log density (-1e29) but a gradient entry that
    overflows (r^2 / scale^3 ~ 1e42).  vi.py:55-58 looks at the density only; here such a sample leaves loss and
    gradients together, on the device step and on the autograd path alike: both equal the oracle's step over the
    other samples, and the variables stay finite."""
    d, s = 3, 8
    model, ds, spec, _ = regression(bvi, gpu_device, 2_000, d, seed=73)
    for fused in (True, False):
        torch.manual_seed(5)
        vi = bvi.VI(model, ds, bvi.ADVI(model, mean_field=mean_field))
        vi.fused_advi = fused
        sur = vi.surrogate_posterior
        scale_var = sur.raw_scale if mean_field else sur.raw_tril
        prm = advi_params(sur)
        eps = f32_noise(np.random.default_rng(74), (s, spec.pu))
        # solve for the noise that puts the unconstrained scale coordinate of draw 5 at log(1e-13)
        if mean_field:
            eps[5, -1] = (math.log(1e-13) - prm['loc'][-1]) / elbo.softplus(prm['raw_scale'][-1])
        else:
            L = elbo.advi_scale_tril(prm['raw_tril'])
            eps[5, -1] = (math.log(1e-13) - prm['loc'][-1] - L[-1, :-1] @ eps[5, :-1].astype(np.float64)) / L[-1, -1]
        inject(sur, [eps], gpu_device)
        params = [p for p in sur.parameters() if p.requires_grad]
        opt = bvi.Adam(learning_rate=0.05).bind(params)
        assert vi._fused_advi_eligible(opt) == fused
        loss = float(vi._step(opt, params, s))
        keep = np.arange(s) != 5
        ref_loss, ref_grads, aux = elbo.advi_loss_and_grads(spec, prm, eps[keep], mean_field)
        assert aux['finite'].all() and np.isfinite(ref_loss)
        ref_loc = elbo.clip_by_norm(ref_grads['loc'])
        ref_scale = elbo.clip_by_norm(ref_grads['raw_scale' if mean_field else 'raw_tril'])
        assert loss == pytest.approx(ref_loss, rel=RTOL)
        assert_grad_close(sur.loc.grad.cpu().numpy()[None], ref_loc[None])
        assert_grad_close(scale_var.grad.cpu().numpy().reshape(1, -1), ref_scale.reshape(1, -1))
        assert bool(torch.isfinite(sur.loc).all()) and bool(torch.isfinite(scale_var).all())


@pytest.mark.parametrize('mean_field', [True, False])
def test_fused_advi_step_equals_the_autograd_step(bvi, gpu_device, mean_field):
    """Mean-field and full-rank ADVI on a model with the built-in priors / bijectors take the step without autograd (sample,
    data pass, closed-form ELBO gradients, clip + Adam in device kernels).  Same loss, clipped gradients, Adam
    moments and variables as the autograd path, step by step -- also with a sample whose log density is not finite
    (dropped from the mean, vi.py:55-58) -- and VI.fit replays it as a graph with the same result."""
    d, s = 5, 16
    model, ds, spec, _ = regression(bvi, gpu_device, 5_000, d, seed=81)
    rng = np.random.default_rng(82)
    noise = [f32_noise(rng, (s, spec.pu)) for _ in range(4)]
    noise[2][3, -1] = -3000.0
    runs = {}
    for fused in (True, False):
        torch.manual_seed(11)
        vi = bvi.VI(model, ds, bvi.ADVI(model, mean_field=mean_field))
        vi.fused_advi = fused
        sur = vi.surrogate_posterior
        scale = sur.raw_scale if mean_field else sur.raw_tril
        inject(sur, noise, gpu_device)
        params = [p for p in sur.parameters() if p.requires_grad]
        opt = bvi.Adam(learning_rate=0.05).bind(params)
        assert vi._fused_advi_eligible(opt) == fused
        rec = []
        for _ in range(4):
            loss = vi._step(opt, params, s)
            rec.append((float(loss), sur.loc.grad.clone(), scale.grad.clone(), sur.loc.detach().clone(),
                        scale.detach().clone(), [m.clone() for m in opt.m], [v.clone() for v in opt.v]))
        assert float(opt.t) == 4.0
        runs[fused] = rec
    for a, b in zip(runs[True], runs[False]):
        assert abs(a[0] - b[0]) <= 2e-6 * abs(b[0])
        for x, y in zip(a[1:5], b[1:5]):
            assert float((x - y).abs().max()) <= 2e-5 * float(y.abs().max()) + 1e-7
        for xs, ys in zip(a[5:], b[5:]):
            for x, y in zip(xs, ys):
                assert float((x - y).abs().max()) <= 2e-5 * float(y.abs().max()) + 1e-9
    # through VI.fit: three eager steps, then graph replay of the fused step
    fits = {}
    for capture in (True, False):
        torch.manual_seed(11)
        vi = bvi.VI(model, ds, bvi.ADVI(model, mean_field=mean_field))
        torch.manual_seed(12)
        _, losses = vi.fit(optimizer=bvi.Adam(learning_rate=0.05), num_steps=12, sample_size=s,
                           num_samples_to_approx_marginals=10, progress_bar=False, capture_graph=capture)
        fits[capture] = (losses, vi.surrogate_posterior.loc.detach().clone())
    assert np.isfinite(fits[True][0]).all() and fits[True][0][-1] < fits[True][0][0]
    np.testing.assert_allclose(fits[True][0], fits[False][0], rtol=1e-5)
    assert float((fits[True][1] - fits[False][1]).abs().max()) <= 1e-5


def test_golden_advi_step_through_the_c_abi(gpu_device):
    """tests/golden/advi_step_small.json (frozen oracle numbers): one mean-field ADVI optimisation step driven
    entry point by entry point through the C ABI -- bvi_advi_meanfield_sample, bvi_constrain, bvi_linreg_logp_grad,
    bvi_posterior_assemble, bvi_advi_meanfield_grads, bvi_clip_adam_step per variable (vi.py:60-75, 109-116).  One of
    the six samples has a log density that is not finite and must drop out of loss and gradients (vi.py:55-58)."""
    import json
    import os
    from bayes_vi_amd import _native as nat
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, 'tests', 'golden', 'advi_step_small.json')) as f:
        gold = json.load(f)
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=gpu_device).contiguous()
    X, y, eps = t(gold['X']), t(gold['y']), t(gold['eps'])
    loc, raw = t(gold['loc']), t(gold['raw_scale'])
    s, pu = eps.shape
    d = X.shape[1]
    blocks = nat.make_blocks([(nat.BIJ_IDENTITY, 1, nat.PRIOR_NORMAL, 0.0, 10.0),
                              (nat.BIJ_IDENTITY, d, nat.PRIOR_NORMAL, 0.0, 10.0),
                              (nat.BIJ_SOFTPLUS, 1, nat.PRIOR_HALFNORMAL, 10.0, 0.0)])
    lib, ptr, st = nat.lib(), nat._ptr, nat._stream()
    f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=gpu_device)
    z, theta, lp, gz = f(s, pu), f(s, pu), f(s), f(s, pu)
    nat.check(lib.bvi_advi_meanfield_sample(s, pu, ptr(eps), ptr(loc), ptr(raw), ptr(z), st), 'sample')
    np.testing.assert_allclose(z.cpu().numpy(), gold['z'], rtol=2e-6, atol=1e-6)
    nat.check(lib.bvi_constrain(blocks, 3, s, ptr(z), ptr(theta), st), 'constrain')
    ll, gl = nat.linreg_logp_grad(X, y, theta, nat.Workspace(gpu_device))
    nat.check(lib.bvi_posterior_assemble(blocks, 3, s, ptr(z), ptr(theta), ptr(ll), ptr(gl), ptr(lp), ptr(gz), st),
              'assemble')
    fin = np.asarray(gold['finite'])
    got_lp = lp.cpu().numpy()
    assert np.isfinite(got_lp).tolist() == fin.tolist()
    assert_logp_close(got_lp[fin], np.asarray(gold['log_post_unconstrained'])[fin])
    assert_grad_close(gz.cpu().numpy()[fin], np.asarray(gold['grad_z'])[fin])
    loss, g_loc, g_raw = f(()), f(pu), f(pu)
    nat.check(lib.bvi_advi_meanfield_grads(s, pu, ptr(lp), ptr(gz), ptr(eps), ptr(raw), ptr(loss), ptr(g_loc),
                                           ptr(g_raw), st), 'grads')
    assert float(loss) == pytest.approx(gold['loss'], rel=RTOL)
    assert_grad_close(g_loc.cpu().numpy()[None], np.asarray(gold['grad']['loc'])[None])
    assert_grad_close(g_raw.cpu().numpy()[None], np.asarray(gold['grad']['raw_scale'])[None])
    step = torch.zeros((), dtype=torch.float64, device=gpu_device)
    lr = gold['learning_rate']
    for i, (x, g, name) in enumerate(((loc, g_loc, 'loc'), (raw, g_raw, 'raw_scale'))):
        m, v = torch.zeros_like(x), torch.zeros_like(x)
        before = x.clone()
        nat.check(lib.bvi_clip_adam_step(pu, ptr(x), ptr(g), ptr(m), ptr(v), ptr(step), 1 if i == 0 else 0, lr, 0.9,
                                         0.999, 1e-7, 1.0, st), 'clip_adam')
        assert_grad_close(g.cpu().numpy()[None], np.asarray(gold['clipped'][name])[None])       # clipped in place
        upd = (x - before).cpu().numpy()
        np.testing.assert_allclose(upd, np.asarray(gold['after_adam'][name]) - np.asarray(gold[name]), rtol=1e-4,
                                   atol=1e-5 * lr)
    assert float(step) == 1.0

