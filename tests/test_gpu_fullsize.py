"""GPU: the hot path at the BASELINE.json sizes (10M x 64 regression with 1024 chains, 100M-row
mixtures), checked through properties that do not need a CPU pass over the data: sums over row
shards add, a rescaling of the problem rescales the answer in closed form, and -- as an independent
full-size reference -- float64 torch evaluations on the device (sufficient statistics for the
Gaussian families, a direct chunked evaluation for the mixture).  Tolerances as in parity_util."""
import math

import numpy as np
import pytest
import torch

from parity_util import RTOL

# gradients are judged against the largest entry of the chain alone: no floor proportional to the summed terms
FLOOR = 0.0
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def nat(gpu_device):
    from bayes_vi_amd import _native
    _native.lib()
    return _native


@pytest.fixture(scope='module')
def ws(nat, gpu_device):
    return nat.Workspace(torch.device(gpu_device))


def _close(a, b, scale, rtol=RTOL, what=''):
    err = (a.double() - b.double()).abs()
    tol = rtol * scale
    assert bool((err <= tol).all()), '%s: worst err/tol %.3g' % (what, float((err / tol).max()))


@pytest.fixture(scope='module')
def regression_10m(gpu_device):
    n, d, c = 10_000_000, 64, 1024
    g = torch.Generator(device=gpu_device); g.manual_seed(1234)
    X = torch.empty(n, d, device=gpu_device)
    for s in range(0, n, 1 << 20):
        X[s:s + (1 << 20)].normal_(generator=g)
    w = torch.randn(d, generator=g, device=gpu_device)
    y = 0.5 + X @ w + torch.randn(n, generator=g, device=gpu_device)
    th = torch.cat([0.5 + 0.01 * torch.randn(c, 1, generator=g, device=gpu_device),
                    w + 0.01 * torch.randn(c, d, generator=g, device=gpu_device),
                    1 + 0.2 * torch.rand(c, 1, generator=g, device=gpu_device)], -1).contiguous()
    # float64 sufficient statistics of the augmented design [1, X]: an independent full-size reference
    A = torch.zeros(d + 1, d + 1, dtype=torch.float64, device=gpu_device)
    b = torch.zeros(d + 1, dtype=torch.float64, device=gpu_device)
    yy = torch.zeros((), dtype=torch.float64, device=gpu_device)
    absx = torch.zeros(d + 1, dtype=torch.float64, device=gpu_device)
    for s in range(0, n, 1 << 20):
        Xa = torch.cat([torch.ones(min(1 << 20, n - s), 1, dtype=torch.float64, device=gpu_device),
                        X[s:s + (1 << 20)].double()], 1)
        yc = y[s:s + (1 << 20)].double()
        A += Xa.T @ Xa; b += Xa.T @ yc; yy += yc @ yc; absx += Xa.abs().sum(0)
    return dict(n=n, d=d, c=c, X=X, y=y, th=th, A=A, b=b, yy=yy, absx=absx)


def _reference_from_stats(r, th):
    n, d = r['n'], r['d']
    beta = th[:, :d + 1].double(); sc = th[:, d + 1].double()
    s2 = r['yy'] - 2 * beta @ r['b'] + ((beta @ r['A']) * beta).sum(-1)
    gb = (r['b'][None] - beta @ r['A']) / sc[:, None] ** 2
    lp = -0.5 * s2 / sc ** 2 - n * torch.log(sc) - 0.5 * n * math.log(2 * math.pi)
    gs = -n / sc + s2 / sc ** 3
    # size of the terms each gradient entry sums: sum_i |r_i| |x_id| ~ sqrt(2/pi) sd(r) sum_i |x_id|
    term = math.sqrt(2 / math.pi) * torch.sqrt(s2 / n)[:, None] * r['absx'][None] / sc[:, None] ** 2
    return lp, torch.cat([gb, gs[:, None]], -1), torch.cat([term, (s2 / sc ** 3 + n / sc)[:, None]], -1)


def test_regression_10m_rows_1024_chains_against_float64_statistics(nat, ws, regression_10m):
    """BASELINE.json configs[1] in full: the f16-piece matrix-core kernel on 10M x 64 with 1024 chains
    against the float64 sufficient-statistics evaluation; the exact-f32 kernel on 64 of the chains."""
    r = regression_10m
    n, d, th = r['n'], r['d'], r['th']
    ref_lp, ref_g, term = _reference_from_stats(r, th)
    pk = nat.linreg_pack(r['X'], r['y'], fmt=nat.PACK_F16)
    lp, g = nat.linreg_packed_logp_grad(pk, n, d, th, ws, fmt=nat.PACK_F16)
    assert bool(torch.isfinite(lp).all()) and bool(torch.isfinite(g).all())
    _close(lp, ref_lp, ref_lp.abs(), what='logp f16 pieces')
    gmax = ref_g.abs().max(-1, keepdim=True).values
    _close(g, ref_g, gmax + FLOOR * term, what='grad f16 pieces')
    lp32, g32 = nat.linreg_logp_grad(r['X'], r['y'], th[:64].contiguous(), ws, precision=nat.PREC_F32)
    _close(lp32, ref_lp[:64], ref_lp[:64].abs(), what='logp f32')
    _close(g32, ref_g[:64], gmax[:64] + FLOOR * term[:64], what='grad f32')
    # determinism at full size
    lp2, g2 = nat.linreg_packed_logp_grad(pk, n, d, th, ws, fmt=nat.PACK_F16)
    assert torch.equal(lp, lp2) and torch.equal(g, g2)
    # value only (random-walk Metropolis): the many-chain kernel leaves out the residuals' low f16 piece, which the
    # log density needs only through a 2^-13-sized compensation term -- same values to float rounding
    lpv, none = nat.linreg_packed_logp_grad(pk, n, d, th, ws, want_grad=False, fmt=nat.PACK_F16)
    assert none is None
    _close(lpv, ref_lp, ref_lp.abs(), what='logp f16 pieces, value only')
    assert float((lpv.double() - lp.double()).abs().max()) <= 4.0      # two float steps at 1e7..1e9
    r['full'] = (lp, g)


@pytest.mark.parametrize('c', [1, 4, 16, 64])
def test_regression_10m_rows_few_chains_through_the_host_mirror(nat, ws, regression_10m, c):
    """From a million rows on the host mirror routes every number of chains to the f16-plane kernel (one chain
    tile per wave, three workgroups per CU): same reference, same bar as for 1024 chains."""
    from bayes_vi_amd.fused_target import LinearRegression
    r = regression_10m
    th = r['th'][100:100 + c].contiguous()
    if 'lik' not in r:
        r['lik'] = LinearRegression(r['X'], r['y'])
    lik = r['lik']
    assert lik.matrix_core_format(c) == nat.PACK_F16
    lp, g = lik.logp_grad(th)
    ref_lp, ref_g, term = _reference_from_stats(r, th)
    _close(lp, ref_lp, ref_lp.abs(), what='logp')
    _close(g, ref_g, ref_g.abs().max(-1, keepdim=True).values + FLOOR * term, what='grad')
    lp_v, _ = lik.logp_grad(th, want_grad=False)
    assert torch.equal(lp, lp_v)


def _level_of(nat, lp):
    """The level bvi_logp_rebase picks for these log-densities (their largest finite value), as a device double."""
    ref = torch.zeros(1, dtype=torch.float64, device=lp.device)
    nat.logp_rebase(lp.clone(), 1, ref)
    return ref


def test_log_density_differences_at_10m_rows(nat, ws, regression_10m):
    """What a Metropolis / NUTS energy test consumes is the DIFFERENCE of two log-densities.  At 10M rows
    |log-likelihood| ~ 1.4e7, where floats are 1 apart, so absolute float outputs carry +-0.5 of noise per decision;
    relative to a level near the chains (logp_ref) the same kernels return differences good to 1e-2.  64 chains
    within the posterior's own width of the mode, against float64 sufficient statistics; the device sampler's
    state through the same check."""
    r = regression_10m
    n, d, dev = r['n'], r['d'], r['X'].device
    g = torch.Generator(device=dev); g.manual_seed(99)
    beta = torch.linalg.solve(r['A'], r['b'])
    sig = torch.sqrt((r['yy'] - beta @ r['b']) / n)
    c = 64
    th = torch.cat([beta[None] + 3e-4 * torch.randn(c, d + 1, generator=g, device=dev, dtype=torch.float64),
                    sig * (1 + 2e-4 * torch.randn(c, 1, generator=g, device=dev, dtype=torch.float64))], -1).float().contiguous()
    ref64, _, _ = _reference_from_stats(r, th)
    assert float(ref64.max() - ref64.min()) < 1e3                 # nearby: the differences are O(parameters)
    pk = nat.linreg_pack(r['X'], r['y'], fmt=nat.PACK_F16)
    lp_abs, _ = nat.linreg_packed_logp_grad(pk, n, d, th, ws, fmt=nat.PACK_F16)
    err_abs = lp_abs.double() - ref64
    level = _level_of(nat, lp_abs)
    th = torch.cat([th, th[:64] + 0.0, th[:64] + 0.0], 0)[:128].contiguous()   # 128 chains: the two-tiles-per-wave kernel
    ref64 = torch.cat([ref64, ref64], 0)[:128]
    for cc in (128, 64, 16):                                       # many-chain (pipelined) and one-tile-per-wave kernels
        lp, _ = nat.linreg_packed_logp_grad(pk, n, d, th[:cc].contiguous(), ws, fmt=nat.PACK_F16, logp_ref=level)
        err = lp.double() + level - ref64[:cc]
        assert float((err - err.mean()).abs().max()) <= 1e-2, float((err - err.mean()).abs().max())
        lpv, _ = nat.linreg_packed_logp_grad(pk, n, d, th[:cc].contiguous(), ws, want_grad=False, fmt=nat.PACK_F16,
                                             logp_ref=level)
        if cc <= 64:
            assert torch.equal(lp, lpv)
        else:                 # value-only form of the many-chain kernel: the residuals' high f16 piece only
            errv = lpv.double() + level - ref64[:cc]
            assert float((errv - errv.mean()).abs().max()) <= 1e-2, float((errv - errv.mean()).abs().max())
    th, ref64 = th[:64].contiguous(), ref64[:64]
    # the absolute float outputs cannot do that (documented, not required): their spread is the float spacing
    assert float((err_abs - err_abs.mean()).abs().max()) > 0.1
    # two row shards, each relative to its own level (half of the total, as bvi_logp_rebase leaves them): they add
    cut = 4_999_937
    half = level / 2
    parts = []
    for lo, hi in ((0, cut), (cut, n)):
        Xs, ys = r['X'][lo:hi], r['y'][lo:hi]
        if Xs.data_ptr() % 16:
            Xs = Xs.clone()
        if ys.data_ptr() % 16:
            ys = ys.clone()
        pks = nat.linreg_pack(Xs, ys, fmt=nat.PACK_F16)
        parts.append(nat.linreg_packed_logp_grad(pks, hi - lo, d, th, ws, fmt=nat.PACK_F16, logp_ref=half)[0])
        del pks
    err = (parts[0] + parts[1]).double() + level - ref64
    assert float((err - err.mean()).abs().max()) <= 1.5e-2
    del pk


def test_device_hmc_keeps_relative_log_densities_at_10m_rows(regression_10m, gpu_device):
    """Device HMC on the 10M-row regression: the state's log-densities are stored relative to a level that
    bvi_logp_rebase moves with the chains (largest current value = 0), the reported target_log_prob is absolute, and
    both agree with float64 -- the relative ones to 1e-2 (plus the priors' float rounding), which absolute floats
    (1 apart at 1.4e7) cannot."""
    from bayes_vi_amd import bijectors as tfb, distributions as tfd
    from bayes_vi_amd.fused_target import LinearRegression, Posterior
    from bayes_vi_amd.inference.mcmc.device_hmc import DeviceHamiltonianMonteCarloKernel
    r = regression_10m
    n, d = r['n'], r['d']
    lik = r.get('lik') or LinearRegression(r['X'], r['y'])
    r['lik'] = lik
    lik.logp_ref = None
    post = Posterior(lik, [(tfd.Normal(0., 10.), 1), (tfd.Normal(torch.zeros(d), 10.), d), (tfd.HalfNormal(10.), 1)],
                     [tfb.Identity(), tfb.Identity(), tfb.Softplus()])
    g = torch.Generator(device=gpu_device); g.manual_seed(3)
    beta = torch.linalg.solve(r['A'], r['b'])
    sig = torch.sqrt((r['yy'] - beta @ r['b']) / n)
    c = 32
    z = torch.cat([beta[None] + 2e-4 * torch.randn(c, d + 1, generator=g, device=gpu_device, dtype=torch.float64),
                   torch.log(torch.expm1(sig)) + 1e-4 * torch.randn(c, 1, generator=g, device=gpu_device,
                                                                    dtype=torch.float64)], -1).float().contiguous()
    kern = DeviceHamiltonianMonteCarloKernel(post, step_size=4e-5, num_leapfrog_steps=4, seed=5)
    assert lik.logp_ref is not None
    st = kern.bootstrap(z)
    accepted = 0
    for _ in range(3):
        z, st, tr = kern.one_step(z, st)
        accepted += int(tr['is_accepted'].sum())
        # float64 log-posterior at the chains' current positions
        theta = torch.cat([z[:, :d + 1], torch.nn.functional.softplus(z[:, d + 1:])], -1)
        ll64, _, _ = _reference_from_stats(r, theta)
        t64 = theta.double()
        prior = (-0.5 * (t64[:, :d + 1] / 10.0) ** 2 - math.log(10.0) - 0.5 * math.log(2 * math.pi)).sum(-1) \
            + (math.log(2.0) - 0.5 * (t64[:, d + 1] / 10.0) ** 2 - math.log(10.0) - 0.5 * math.log(2 * math.pi))
        fldj = torch.log(torch.sigmoid(z[:, d + 1].double()))
        ref = ll64 + prior + fldj
        rel, lvl = st['relative_log_prob'], st['log_prob_level']
        assert float(rel.max()) == 0.0                             # re-levelled to the best chain
        err = rel.double() + lvl - ref
        assert float((err - err.mean()).abs().max()) <= 2e-2, float((err - err.mean()).abs().max())
        assert float((st['target_log_prob'].double() - ref).abs().max()) <= 2.0      # absolute: float spacing
        assert torch.equal(tr['target_log_prob'], st['target_log_prob'])
    assert accepted > 0
    # the level is the samplers' business only: every other evaluation of the same posterior stays absolute
    lp_abs, _ = post.unconstrained_log_prob_and_grad(z)
    assert float((lp_abs.double() - ref).abs().max()) <= 2.0
    assert float(post.log_prob(theta).double().sub(ref - fldj).abs().max()) <= 2.0


def test_regression_10m_row_shards_add(nat, ws, regression_10m):
    """The outputs are likelihood-only sums, so two row shards -- packed separately, with their own
    column scales -- must add up to the full-data result (what the all-reduce over GPUs relies on)."""
    r = regression_10m
    n, d, th = r['n'], r['d'], r['th']
    if 'full' not in r:
        pk = nat.linreg_pack(r['X'], r['y'], fmt=nat.PACK_F16)
        r['full'] = nat.linreg_packed_logp_grad(pk, n, d, th, ws, fmt=nat.PACK_F16)
    lp, g = r['full']
    cut = 3_700_001
    parts = []
    for lo, hi in ((0, cut), (cut, n)):
        Xs, ys = r['X'][lo:hi], r['y'][lo:hi]
        pk = nat.linreg_pack(Xs, ys, fmt=nat.PACK_F16)
        parts.append(nat.linreg_packed_logp_grad(pk, hi - lo, d, th, ws, fmt=nat.PACK_F16))
        del pk
    lps = parts[0][0].double() + parts[1][0].double()
    gs = parts[0][1].double() + parts[1][1].double()
    _, ref_g, term = _reference_from_stats(r, th)
    _close(lps, lp, lp.abs(), what='logp shards')
    _close(gs, g, ref_g.abs().max(-1, keepdim=True).values + FLOOR * term, what='grad shards')


def test_regression_10m_rescaling_property(nat, ws, regression_10m):
    """y -> a y, (b0, w, scale) -> a (b0, w, scale) leaves every standardised residual unchanged:
    logp shifts by -n log a and the gradient scales by 1/a (a = 3: not a power of two, so every
    operand's bits change)."""
    r = regression_10m
    n, d, th = r['n'], r['d'], r['th'][:256].contiguous()
    a = 3.0
    pk = nat.linreg_pack(r['X'], r['y'], fmt=nat.PACK_F16)
    lp, g = nat.linreg_packed_logp_grad(pk, n, d, th, ws, fmt=nat.PACK_F16)
    del pk
    ya = (r['y'] * a).contiguous()
    pka = nat.linreg_pack(r['X'], ya, fmt=nat.PACK_F16)
    lpa, ga = nat.linreg_packed_logp_grad(pka, n, d, (th * a).contiguous(), ws, fmt=nat.PACK_F16)
    _, ref_g, term = _reference_from_stats(r, r['th'][:256])
    _close(lpa.double() + n * math.log(a), lp, lp.abs(), rtol=2 * RTOL, what='logp rescaled')
    # y, theta were re-rounded to f32 after scaling: allow the f32 input rounding on top (2x)
    _close(ga.double() * a, g, ref_g.abs().max(-1, keepdim=True).values + FLOOR * term, rtol=3 * RTOL,
           what='grad rescaled')


def _mixture_float64(y, th, k):
    """Chunked float64 evaluation of the 1-D mixture on the device: log density, gradient and the sum of
    absolute per-row gradient terms (the scale rounding errors are measured against) for a few chains."""
    n, dev = y.shape[0], y.device
    t2 = th.double(); c = t2.shape[0]
    wgt, mu, sc = t2[:, :k], t2[:, k:2 * k], t2[:, 2 * k:]
    ref_lp = torch.zeros(c, dtype=torch.float64, device=dev)
    ref_g = torch.zeros(c, 3 * k, dtype=torch.float64, device=dev)
    absg = torch.zeros(c, 3 * k, dtype=torch.float64, device=dev)
    logw = torch.log(wgt / wgt.sum(-1, keepdim=True))
    step = max(1, (1 << 25) // (c * k))
    for s in range(0, n, step):
        yc = y[s:s + step].double()[None, :, None]
        u = (yc - mu[:, None]) / sc[:, None]
        e = logw[:, None] - torch.log(sc[:, None]) - 0.5 * math.log(2 * math.pi) - 0.5 * u * u
        lse = torch.logsumexp(e, -1)
        rho = torch.exp(e - lse[..., None])
        ref_lp += lse.sum(-1)
        parts = [rho / wgt[:, None] - 1.0 / wgt.sum(-1)[:, None, None], rho * u / sc[:, None],
                 rho * (u * u - 1) / sc[:, None]]
        ref_g += torch.cat([p.sum(1) for p in parts], -1)
        absg += torch.cat([p.abs().sum(1) for p in parts], -1)
    return ref_lp, ref_g, absg


@pytest.mark.parametrize('n', [3_000_001, 20_000_003])
@pytest.mark.parametrize('case', ['separated', 'overlapping', 'k3', 'k16'])
def test_mixture_component_table(nat, ws, gpu_device, n, case):
    """Inputs long enough for the kernel to build its table of the components that matter at each y (128 bins at
    3M rows, 1024 at 20M): separated and heavily overlapping components, very unequal weights, K below and at
    the template sizes, rows far outside every component (the per-row-maximum path), one chain with a NaN mean.
    Chains 0-2 and the last one against float64; the value-only launch returns the same log densities."""
    g = torch.Generator(device=gpu_device); g.manual_seed(11)
    k = {'separated': 8, 'overlapping': 6, 'k3': 3, 'k16': 16}[case]
    c = 64 if case != 'k16' else 9          # 9 chains of K = 16: the 16-chain shuffle mapping
    if case == 'overlapping':
        locs = torch.linspace(-2, 2, k, device=gpu_device)
        wbase = torch.tensor([0.01, 0.6, 0.05, 0.2, 0.04, 0.1], device=gpu_device)
        sbase = torch.tensor([0.2, 1.0, 5.0, 0.5, 2.0, 0.3], device=gpu_device)
    else:
        locs = torch.linspace(-20, 20, k, device=gpu_device)
        wbase = torch.full((k,), 1.0 / k, device=gpu_device)
        sbase = torch.ones(k, device=gpu_device)
    comp = torch.multinomial(wbase, n, replacement=True, generator=g)
    y = locs[comp] + sbase[comp] * torch.randn(n, generator=g, device=gpu_device)
    del comp
    y[::100_003] = 1000.0            # rows no component reaches
    y[7::250_007] = -37.5
    y[11::300_017] = 2.857           # between two of the separated components
    th = torch.cat([wbase * (1 + 0.1 * torch.rand(c, k, generator=g, device=gpu_device)),
                    locs + 0.1 * torch.randn(c, k, generator=g, device=gpu_device),
                    sbase * (1 + 0.1 * torch.rand(c, k, generator=g, device=gpu_device))], -1).contiguous()
    th[c - 1, k:2 * k] += 3.0        # one chain far from the others: its components widen the wave's table entries
    lp, gr = nat.gmm1d_logp_grad(y, th, k, ws)
    lpv, _ = nat.gmm1d_logp_grad(y, th, k, ws, want_grad=False)
    assert torch.equal(lp, lpv)
    sel = torch.tensor([0, 1, 2, c - 1], device=gpu_device)
    ref_lp, ref_g, absg = _mixture_float64(y, th[sel], k)
    _close(lp[sel], ref_lp, ref_lp.abs(), what='mixture logp (%s)' % case)
    _close(gr[sel], ref_g, ref_g.abs().max(-1, keepdim=True).values + FLOOR * absg, what='mixture grad (%s)' % case)
    bad = th.clone(); bad[1, k + 1] = float('nan')
    lpb, grb = nat.gmm1d_logp_grad(y, bad, k, ws)
    assert bool(torch.isnan(lpb[1])) and bool(torch.isfinite(lpb[[0, 2]]).all())
    assert torch.equal(lpb[[0, 2, c - 1]], lp[[0, 2, c - 1]])


@pytest.mark.parametrize('c', [1, 2, 3, 4])
@pytest.mark.parametrize('case', ['separated', 'overlapping', 'k16'])
def test_mixture_few_chains_rows_in_ascending_order(nat, ws, gpu_device, c, case):
    """The few-chain kernel (lanes hold rows) leaves a component out for the 256 rows a wave has loaded when all of
    them are out of its reach, which happens when the rows are stored in ascending order (what the host layer does).
    Same inputs in ascending and in the original order, against float64: separated components (most are left out),
    overlapping ones (none is), K = 16 (one chain per group), a stretch of rows no component reaches, a NaN row's
    worth of propagation, one chain with a NaN mean; the value-only launch returns the same log densities.  Two to
    four chains with the gradient run as groups of two chains (the last group of three chains holds one)."""
    n = 5_000_011
    g = torch.Generator(device=gpu_device); g.manual_seed(23)
    k = {'separated': 8, 'overlapping': 6, 'k16': 16}[case]
    if case == 'overlapping':
        locs = torch.linspace(-2, 2, k, device=gpu_device)
        wbase = torch.tensor([0.01, 0.6, 0.05, 0.2, 0.04, 0.1], device=gpu_device)
        sbase = torch.tensor([0.2, 1.0, 5.0, 0.5, 2.0, 0.3], device=gpu_device)
    else:
        locs = torch.linspace(-20, 20, k, device=gpu_device)
        wbase = torch.full((k,), 1.0 / k, device=gpu_device)
        sbase = torch.ones(k, device=gpu_device)
    comp = torch.multinomial(wbase, n, replacement=True, generator=g)
    y = locs[comp] + sbase[comp] * torch.randn(n, generator=g, device=gpu_device)
    del comp
    y[::100_003] = 1000.0            # rows no component reaches: a stretch of them at the upper end once sorted
    y[11::300_017] = 2.857
    th = torch.cat([wbase * (1 + 0.1 * torch.rand(c, k, generator=g, device=gpu_device)),
                    locs + 0.1 * torch.randn(c, k, generator=g, device=gpu_device),
                    sbase * (1 + 0.1 * torch.rand(c, k, generator=g, device=gpu_device))], -1).contiguous()
    ys = torch.sort(y).values.contiguous()
    ref_lp, ref_g, absg = _mixture_float64(ys, th, k)
    for rows, what in ((ys, 'ascending'), (y, 'as drawn')):
        lp, gr = nat.gmm1d_logp_grad(rows, th, k, ws)
        lpv, _ = nat.gmm1d_logp_grad(rows, th, k, ws, want_grad=False)
        _close(lp, ref_lp, ref_lp.abs(), what='mixture logp, %d chains, rows %s (%s)' % (c, what, case))
        _close(lpv, ref_lp, ref_lp.abs(), what='mixture logp, value only, %d chains, rows %s (%s)' % (c, what, case))
        _close(gr, ref_g, ref_g.abs().max(-1, keepdim=True).values + FLOOR * absg,
               what='mixture grad, %d chains, rows %s (%s)' % (c, what, case))
    # an odd start (the 16-byte grid of the loads shifts) and non-finite inputs
    lpo, _ = nat.gmm1d_logp_grad(ys[1:], th, k, ws, want_grad=False)
    ref_o, _, _ = _mixture_float64(ys[1:], th, k)
    _close(lpo, ref_o, ref_o.abs(), what='mixture logp, odd start')
    bad = th.clone(); bad[0, k + 1] = float('nan')
    lpb, grb = nat.gmm1d_logp_grad(ys, bad, k, ws)
    assert bool(torch.isnan(lpb[0])) and bool(torch.isfinite(lpb[1:]).all())
    ynan = ys.clone(); ynan[n // 2] = float('nan')
    lpn, _ = nat.gmm1d_logp_grad(ynan, th, k, ws, want_grad=False)
    assert bool(torch.isnan(lpn).all())


def test_mixture_100m_rows(nat, ws, gpu_device):
    """BASELINE.json configs[2] size: K = 8 mixture over 100M rows.  Row shards (cut at an odd row, so the
    second shard is not 16-byte aligned) add up; two chains against a chunked float64 evaluation."""
    n, k, c = 100_000_000, 8, 16
    g = torch.Generator(device=gpu_device); g.manual_seed(7)
    locs = torch.linspace(-20, 20, k, device=gpu_device)
    y = torch.empty(n, device=gpu_device)
    for s in range(0, n, 1 << 24):
        m = min(1 << 24, n - s)
        y[s:s + m] = locs[torch.randint(0, k, (m,), generator=g, device=gpu_device)] + \
            torch.randn(m, generator=g, device=gpu_device)
    th = torch.cat([(1.0 / k) * (1 + 0.1 * torch.rand(c, k, generator=g, device=gpu_device)),
                    locs + 0.1 * torch.randn(c, k, generator=g, device=gpu_device),
                    1 + 0.1 * torch.rand(c, k, generator=g, device=gpu_device)], -1).contiguous()
    lp, gr = nat.gmm1d_logp_grad(y, th, k, ws)
    cut = 41_234_567
    a = nat.gmm1d_logp_grad(y[:cut], th, k, ws)
    b = nat.gmm1d_logp_grad(y[cut:], th, k, ws)
    ref_lp, ref_g, absg = _mixture_float64(y, th[:2], k)
    _close(lp[:2], ref_lp, ref_lp.abs(), what='mixture logp')
    _close(gr[:2], ref_g, ref_g.abs().max(-1, keepdim=True).values + FLOOR * absg, what='mixture grad')
    _close(a[0].double() + b[0].double(), lp, lp.abs(), what='mixture logp shards')
    scale = gr.abs().max(-1, keepdim=True).values.double() + FLOOR * absg.max(0).values[None]
    _close(a[1].double() + b[1].double(), gr, scale, what='mixture grad shards')
    # differences between nearby chains, relative to a level: |logp| ~ 3.5e8, floats 32 apart.  The kernel's own
    # exp2 / log2 leave a common offset of a few nats (1e-8 of the total) that cancels between chains; what a
    # Metropolis test sees is the spread.  (Parity unpinned for this family: the reference holds no fixture.)
    th = torch.cat([(1.0 / k) * (1 + 1e-4 * torch.randn(64, k, generator=g, device=gpu_device)),
                    locs + 1e-4 * torch.randn(64, k, generator=g, device=gpu_device),
                    1 + 1e-4 * torch.randn(64, k, generator=g, device=gpu_device)], -1).contiguous()
    ref64, _, _ = _mixture_float64(y, th[:6], k)
    level = _level_of(nat, nat.gmm1d_logp_grad(y, th, k, ws, want_grad=False)[0])
    for want_grad in (True, False):
        rel, _ = nat.gmm1d_logp_grad(y, th, k, ws, want_grad=want_grad, logp_ref=level)
        err = rel[:6].double() + level - ref64
        assert float(err.abs().max()) <= 10.0 and float((err - err.mean()).abs().max()) <= 6e-2, err.tolist()


def test_normal_100m_rows(nat, ws, gpu_device):
    n, c = 100_000_000, 64
    g = torch.Generator(device=gpu_device); g.manual_seed(9)
    y = torch.empty(n, device=gpu_device).normal_(7.0, 3.0, generator=g)
    th = torch.stack([7 + 0.01 * torch.randn(c, generator=g, device=gpu_device),
                      3 + 0.1 * torch.rand(c, generator=g, device=gpu_device)], -1).contiguous()
    lp, gr = nat.normal_iid_logp_grad(y, th, ws)
    s0 = torch.zeros((), dtype=torch.float64, device=gpu_device); s1 = s0.clone(); s2 = s0.clone(); sa = s0.clone()
    for s in range(0, n, 1 << 24):
        yc = y[s:s + (1 << 24)].double()
        s1 += yc.sum(); s2 += (yc * yc).sum(); sa += (yc - 7.0).abs().sum()
    mu, sc = th[:, 0].double(), th[:, 1].double()
    r1 = s1 - n * mu
    r2 = s2 - 2 * mu * s1 + n * mu * mu
    ref_lp = -0.5 * r2 / sc ** 2 - n * torch.log(sc) - 0.5 * n * math.log(2 * math.pi)
    ref_g = torch.stack([r1 / sc ** 2, -n / sc + r2 / sc ** 3], -1)
    term = torch.stack([sa / sc ** 2, r2 / sc ** 3 + n / sc], -1)
    _close(lp, ref_lp, ref_lp.abs(), what='normal logp')
    _close(gr, ref_g, ref_g.abs().max(-1, keepdim=True).values + FLOOR * term, what='normal grad')
    # differences between chains inside the posterior's width, relative to a level: |logp| ~ 2.5e8 (floats 16 apart)
    mu_hat = s1 / n; sd_hat = torch.sqrt(s2 / n - mu_hat ** 2)
    for cc, bar in ((64, 2e-2), (16, 0.5), (1, 0.5)):      # few chains: one float sum of ~400 rows per lane
        th = torch.stack([mu_hat + 3e-4 * torch.randn(cc, generator=g, device=gpu_device, dtype=torch.float64),
                          sd_hat * (1 + 1e-4 * torch.randn(cc, generator=g, device=gpu_device, dtype=torch.float64))],
                         -1).float().contiguous()
        mu, sc = th[:, 0].double(), th[:, 1].double()
        ref64 = -0.5 * (s2 - 2 * mu * s1 + n * mu * mu) / sc ** 2 - n * torch.log(sc) - 0.5 * n * math.log(2 * math.pi)
        level = _level_of(nat, nat.normal_iid_logp_grad(y, th, ws)[0])
        rel, _ = nat.normal_iid_logp_grad(y, th, ws, logp_ref=level)
        err = rel.double() + level - ref64
        assert float(err.abs().max()) <= bar, (cc, float(err.abs().max()))


def test_fullcov_mixture_1m_rows_shards_add(nat, ws, gpu_device):
    """BASELINE.json configs[3] shape (D = 32, K = 16, full covariance): shard additivity at 1M rows."""
    n, d, k, c = 1_000_000, 32, 16, 4
    g = torch.Generator(device=gpu_device); g.manual_seed(11)
    centers = 4 * torch.randn(k, d, generator=g, device=gpu_device)
    Y = (centers[torch.randint(0, k, (n,), generator=g, device=gpu_device)]
         + torch.randn(n, d, generator=g, device=gpu_device)).contiguous()
    L = torch.eye(d, device=gpu_device)[None, None] * (1 + 0.1 * torch.rand(c, k, 1, 1, generator=g, device=gpu_device)) \
        + 0.02 * torch.tril(torch.randn(c, k, d, d, generator=g, device=gpu_device), -1)
    th = torch.cat([(1.0 / k) * (1 + 0.1 * torch.rand(c, k, generator=g, device=gpu_device)),
                    (centers[None] + 0.05 * torch.randn(c, k, d, generator=g, device=gpu_device)).reshape(c, -1),
                    L.reshape(c, -1)], -1).contiguous()
    lp, gr = nat.gmm_fullcov_logp_grad(Y, th, k, ws)
    cut = 333_337
    a = nat.gmm_fullcov_logp_grad(Y[:cut].contiguous(), th, k, ws)
    b = nat.gmm_fullcov_logp_grad(Y[cut:].contiguous(), th, k, ws)
    assert bool(torch.isfinite(lp).all()) and bool(torch.isfinite(gr).all())
    _close(a[0].double() + b[0].double(), lp, lp.abs(), what='fullcov logp shards')
    # per-entry term scale is not available without a second pass: judge against the largest entry of the
    # chain's gradient block (weights | locs | scale_tril)
    for lo, hi in ((0, k), (k, k + k * d), (k + k * d, th.shape[1])):
        scale = gr[:, lo:hi].abs().max(-1, keepdim=True).values.double()
        _close((a[1].double() + b[1].double())[:, lo:hi], gr[:, lo:hi], scale, rtol=3 * RTOL, what='fullcov grad shards')
    # the value-only launch (no responsibilities, no raw sums) returns the same log-densities
    lpv, none = nat.gmm_fullcov_logp_grad(Y, th, k, ws, want_grad=False)
    assert none is None and torch.equal(lp, lpv)
    # float64 evaluation of the first 200k rows for two chains; then differences relative to a level
    # (|logp| ~ 1e7 per 200k rows; parity unpinned for this family: the reference holds no fixture for it)
    m = 200_000
    Ym = Y[:m].contiguous()
    t64 = th[:2].double()
    w = t64[:, :k]; mu = t64[:, k:k + k * d].reshape(2, k, d); Lc = torch.tril(t64[:, k + k * d:].reshape(2, k, d, d))
    ref64 = torch.zeros(2, dtype=torch.float64, device=gpu_device)
    for s0 in range(0, m, 20_000):
        yc = Ym[s0:s0 + 20_000].double()
        diff = yc[None, None] - mu[:, :, None]                                   # [2, k, rows, d]
        z = torch.linalg.solve_triangular(Lc, diff.transpose(-1, -2), upper=False)   # [2, k, d, rows]
        e = torch.log(w / w.sum(-1, keepdim=True))[..., None] - torch.log(torch.diagonal(Lc, dim1=-2, dim2=-1)).sum(-1)[..., None] \
            - 0.5 * d * math.log(2 * math.pi) - 0.5 * (z * z).sum(-2)
        ref64 += torch.logsumexp(e, 1).sum(-1)
    lpm, _ = nat.gmm_fullcov_logp_grad(Ym, th[:2].contiguous(), k, ws)
    _close(lpm, ref64, ref64.abs(), what='fullcov logp 200k rows')
    level = _level_of(nat, lpm)
    rel, _ = nat.gmm_fullcov_logp_grad(Ym, th[:2].contiguous(), k, ws, logp_ref=level)
    err = rel.double() + level - ref64
    assert float(err.abs().max()) <= 0.5, err.tolist()


def test_fullcov_mixture_1m_rows_order_and_pack_do_not_matter(nat, ws, gpu_device):
    """configs[3] shape at 1M rows: the rows as drawn through bvi_gmm_fullcov_logp_grad against the rows stored cluster by
    cluster (bvi_cluster_rows) and packed (bvi_gmm_fullcov_pack) through bvi_gmm_fullcov_packed_logp_grad -- the
    dense kernel against the pruned kernels on the same sums.  A sum over rows does not depend on their order: the
    two must agree to float rounding of the sums, for well separated and for overlapping components, and the ordering
    must be a permutation (every row once: compared through order-independent checksums of the rows)."""
    n, d, k, c = 1_000_000, 32, 16, 6
    for spread in (4.0, 0.8):                       # centres 4 widths apart per coordinate; 0.8: clusters overlap
        g = torch.Generator(device=gpu_device); g.manual_seed(int(10 * spread))
        centers = spread * torch.randn(k, d, generator=g, device=gpu_device)
        Y = (centers[torch.randint(0, k, (n,), generator=g, device=gpu_device)]
             + torch.randn(n, d, generator=g, device=gpu_device)).contiguous()
        L = torch.eye(d, device=gpu_device)[None, None] * (1 + 0.1 * torch.rand(c, k, 1, 1, generator=g, device=gpu_device)) \
            + 0.02 * torch.tril(torch.randn(c, k, d, d, generator=g, device=gpu_device), -1)
        th = torch.cat([(1.0 / k) * (1 + 0.1 * torch.rand(c, k, generator=g, device=gpu_device)),
                        (centers[None] + 0.05 * torch.randn(c, k, d, generator=g, device=gpu_device)).reshape(c, -1),
                        L.reshape(c, -1)], -1).contiguous()
        Yc, order = nat.cluster_rows(Y, 64, want_order=True)
        assert torch.equal(torch.sort(order).values, torch.arange(n, device=gpu_device))
        assert torch.equal(Yc.view(torch.int32).sum(0), Y.view(torch.int32).sum(0))      # integer sums: exact
        packed = nat.gmm_fullcov_pack(Yc)
        lp, gr = nat.gmm_fullcov_logp_grad(Y, th, k, ws)
        lp2, gr2 = nat.gmm_fullcov_logp_grad(Yc, th, k, ws, packed=packed)
        assert bool(torch.isfinite(lp2).all()) and bool(torch.isfinite(gr2).all())
        _close(lp2, lp, lp.abs(), what='fullcov logp, rows reordered (spread %g)' % spread)
        for lo, hi in ((0, k), (k, k + k * d), (k + k * d, th.shape[1])):
            scale = gr[:, lo:hi].abs().max(-1, keepdim=True).values.double()
            _close(gr2[:, lo:hi], gr[:, lo:hi], scale, rtol=3 * RTOL, what='fullcov grad, rows reordered (spread %g)' % spread)


# ---- BASELINE.json configs[4]: ELBO gradients on the 10M x 64 regression with 256 Monte-Carlo samples ----
def _vi_setup(r, gpu_device, rows=None):
    import collections
    import bayes_vi_amd.distributions as tfd
    from bayes_vi_amd import symbolic
    from bayes_vi_amd.inference.vi import VI
    from bayes_vi_amd.inference.vi.surrogate_posteriors import ADVI
    from bayes_vi_amd.model import Model
    from bayes_vi_amd.utils.datasets import TensorDataset
    d = r['d']
    priors = collections.OrderedDict(b=tfd.Normal(0., 10.), w=tfd.Normal(torch.zeros(d), 10.), scale=tfd.HalfNormal(10.))
    model = Model(priors, lambda b, w, scale, features: tfd.Normal(loc=b + symbolic.matvec(features, w), scale=scale))
    lo, hi = rows if rows is not None else (0, r['n'])
    vi = VI(model, TensorDataset(r['X'][lo:hi], r['y'][lo:hi]), ADVI(model, mean_field=True))
    return model, vi


def _set_surrogate(sur, loc, raw_scale, eps):
    with torch.no_grad():
        sur.loc.copy_(loc); sur.raw_scale.copy_(raw_scale)
    sur.loc.grad = None; sur.raw_scale.grad = None
    sur.base_noise = lambda n, dim: eps


@pytest.mark.parametrize('regime', ['initial', 'converged'])
def test_elbo_gradient_10m_rows_256_samples(regression_10m, gpu_device, regime):
    """configs[4] at full size through VI.loss: the mean-field ADVI ELBO and its gradient for 256 samples
    (vi.py:109-116, surrogate_posteriors.py:113-123) against (i) a float64 evaluation from the sufficient
    statistics, with the ELBO's chain rule in closed form, and (ii) the sum over two row shards (what the
    all-reduce of partial log-likelihoods relies on).  'initial': the reference's starting surrogate (loc ~ N(0, 1),
    unit scale); 'converged': centred on the data-generating parameters with the posterior's own width, where the
    gradient is a small difference of large sums."""
    r = regression_10m
    n, d, S = r['n'], r['d'], 256
    pu = d + 2
    g = torch.Generator(device=gpu_device); g.manual_seed(99)
    eps = torch.randn(S, pu, generator=g, device=gpu_device)
    if regime == 'initial':
        loc = torch.randn(pu, generator=g, device=gpu_device)
        raw = torch.full((pu,), math.log(math.e - 1.0), device=gpu_device)
    else:
        beta = torch.linalg.solve(r['A'], r['b'])                                  # least squares, float64
        s2 = float(r['yy'] - beta @ r['b']) / n
        loc = torch.cat([beta, torch.tensor([math.log(math.expm1(math.sqrt(s2)))], device=gpu_device,
                                            dtype=torch.float64)]).float()
        sd = math.sqrt(s2 / n)
        raw = torch.full((pu,), math.log(math.expm1(sd)), device=gpu_device)
    model, vi = _vi_setup(r, gpu_device)
    sur = vi.surrogate_posterior
    _set_surrogate(sur, loc, raw, eps)
    loss = vi.loss(S)
    loss.backward()
    # float64 reference: per-sample constrained log-likelihood and gradient from the sufficient statistics
    sc = torch.nn.functional.softplus(raw.double())
    z = loc.double() + sc * eps.double()
    theta = torch.cat([z[:, :d + 1], torch.nn.functional.softplus(z[:, d + 1:])], -1)
    ll, gth, term = _reference_from_stats(r, theta.float())     # evaluated at the f32 parameters the kernel sees
    th32 = theta.float().double()
    lprior = (-0.5 * (th32[:, :d + 1] / 10) ** 2 - 0.5 * math.log(2 * math.pi) - math.log(10)).sum(-1) \
        + 0.5 * math.log(2 / math.pi) - math.log(10) - 0.5 * (th32[:, d + 1] / 10) ** 2
    gprior = -th32 / 100.0
    zs = z[:, d + 1]
    fldj = -torch.nn.functional.softplus(-zs)
    logq = (-0.5 * eps.double() ** 2 - 0.5 * math.log(2 * math.pi) - torch.log(sc)).sum(-1) - fldj
    ref_loss = (logq - ll - lprior).mean()
    gz = gth + gprior
    gz[:, d + 1] = gz[:, d + 1] * torch.sigmoid(zs) + torch.sigmoid(-zs)
    ref_gloc = -gz.mean(0)
    ref_graw = -((1.0 / sc + gz * eps.double()).mean(0)) * torch.sigmoid(raw.double())
    # error scale of a gradient entry: that of the per-sample likelihood gradients it averages
    tz = term.clone(); tz[:, d + 1] = tz[:, d + 1] * torch.sigmoid(zs)
    gmax = gth.abs().max(-1, keepdim=True).values
    per_sample = RTOL * (gmax + FLOOR * tz)
    tol_loc = per_sample.mean(0)
    tol_raw = (per_sample * eps.double().abs()).mean(0) * torch.sigmoid(raw.double())
    assert abs(float(loss.detach()) - float(ref_loss)) <= RTOL * abs(float(ref_loss))
    assert bool(((sur.loc.grad.double() - ref_gloc).abs() <= tol_loc + RTOL * ref_gloc.abs().max()).all()), \
        float(((sur.loc.grad.double() - ref_gloc).abs() / (tol_loc + RTOL * ref_gloc.abs().max())).max())
    assert bool(((sur.raw_scale.grad.double() - ref_graw).abs() <= tol_raw + RTOL * ref_graw.abs().max()).all()), \
        float(((sur.raw_scale.grad.double() - ref_graw).abs() / (tol_raw + RTOL * ref_graw.abs().max())).max())
    # row shards: loss and gradient are sums of per-shard parts once the data-free part is counted once
    full = (loss.detach().double(), sur.loc.grad.double().clone(), sur.raw_scale.grad.double().clone())
    del vi
    cut = 6_100_003
    acc = [torch.zeros((), dtype=torch.float64, device=gpu_device), torch.zeros(pu, dtype=torch.float64, device=gpu_device),
           torch.zeros(pu, dtype=torch.float64, device=gpu_device)]
    for rows in ((0, cut), (cut, n)):
        _, vs = _vi_setup(r, gpu_device, rows)
        _set_surrogate(vs.surrogate_posterior, loc, raw, eps)
        ls = vs.loss(S); ls.backward()
        acc[0] += ls.detach().double(); acc[1] += vs.surrogate_posterior.loc.grad.double()
        acc[2] += vs.surrogate_posterior.raw_scale.grad.double()
        del vs
    # the data-free part (log q - log prior), counted twice above
    lo_ = loc.clone().requires_grad_(True); ra_ = raw.clone().requires_grad_(True)
    sc32 = torch.nn.functional.softplus(ra_)
    z32 = lo_ + sc32 * eps
    th = torch.cat([z32[:, :d + 1], torch.nn.functional.softplus(z32[:, d + 1:])], -1)
    lq = (-0.5 * eps ** 2 - 0.5 * math.log(2 * math.pi) - torch.log(sc32)).sum(-1) + torch.nn.functional.softplus(-z32[:, d + 1])
    lpr = (-0.5 * (th[:, :d + 1] / 10) ** 2 - 0.5 * math.log(2 * math.pi) - math.log(10)).sum(-1) \
        + 0.5 * math.log(2 / math.pi) - math.log(10) - 0.5 * (th[:, d + 1] / 10) ** 2
    l0 = (lq - lpr).mean(); l0.backward()
    assert abs(float(acc[0] - l0.detach().double() - full[0])) <= 2 * RTOL * abs(float(full[0]))
    assert bool(((acc[1] - lo_.grad.double() - full[1]).abs() <= 2 * (tol_loc + RTOL * ref_gloc.abs().max())).all())
    assert bool(((acc[2] - ra_.grad.double() - full[2]).abs() <= 2 * (tol_raw + RTOL * ref_graw.abs().max())).all())


def test_vi_step_10m_rows_graph_replay_equals_eager(regression_10m, gpu_device):
    """VI.fit replays the optimisation step as a HIP graph (sampling, fused pass, backward, clipping, Adam): one
    replayed step from a given state must leave exactly the variables an eagerly executed step leaves."""
    from bayes_vi_amd.inference.vi.vi import Adam
    r = regression_10m
    S, pu = 256, r['d'] + 2
    g = torch.Generator(device=gpu_device); g.manual_seed(5)
    eps = torch.randn(S, pu, generator=g, device=gpu_device)
    loc0 = torch.randn(pu, generator=g, device=gpu_device)
    raw0 = torch.full((pu,), math.log(math.e - 1.0), device=gpu_device)
    _, vi = _vi_setup(r, gpu_device)
    sur = vi.surrogate_posterior
    params = [p for p in sur.parameters() if p.requires_grad]
    _set_surrogate(sur, loc0, raw0, eps)
    sur.loc.grad = torch.zeros_like(sur.loc); sur.raw_scale.grad = torch.zeros_like(sur.raw_scale)
    opt = Adam(learning_rate=0.01).bind(params)
    loss_e = vi._step(opt, params, S).clone()
    eager = [p.detach().clone() for p in params]
    assert not torch.equal(eager[0], loc0)
    # record the same step, then replay it from the same state with a fresh optimizer state
    opt2 = Adam(learning_rate=0.01).bind(params)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = vi._step(opt2, params, S)
    with torch.no_grad():
        sur.loc.copy_(loc0); sur.raw_scale.copy_(raw0)
        opt2.t.zero_()
        for t in opt2.m + opt2.v:
            t.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_loss, loss_e)
    for p, e in zip(params, eager):
        assert torch.equal(p.detach(), e)


# ---- configs[4], second half: the Hamiltonian normalizing flow on the same 10M-row regression ----
def _hnf_setup(r, gpu_device, sur=None, rows=None):
    """VI over NormalizingFlow(Chain of two HamiltonianFlow bijectors, 2 leapfrog steps, hidden [128, 128]) with Pu
    extra dimensions and an MLP lift distribution: linear-regression-example.ipynb cell 13 on the D = 64 model
    (flow_bijectors.py:15-85, surrogate_posteriors.py:153-212)."""
    import collections
    import bayes_vi_amd.distributions as tfd
    from bayes_vi_amd import symbolic
    from bayes_vi_amd.inference.vi import VI
    from bayes_vi_amd.inference.vi.flow_bijectors import Chain, HamiltonianFlow, make_scale_fn, make_shift_fn
    from bayes_vi_amd.inference.vi.surrogate_posteriors import NormalizingFlow
    from bayes_vi_amd.model import Model
    from bayes_vi_amd.utils.datasets import TensorDataset
    d = r['d']
    priors = collections.OrderedDict(b=tfd.Normal(0., 10.), w=tfd.Normal(torch.zeros(d), 10.), scale=tfd.HalfNormal(10.))
    model = Model(priors, lambda b, w, scale, features: tfd.Normal(loc=b + symbolic.matvec(features, w), scale=scale))
    if sur is None:
        nd = model.flat_unconstrained_param_event_ndims
        torch.manual_seed(17)
        flows = Chain([HamiltonianFlow(event_dims=nd, step_sizes=torch.nn.Parameter(torch.tensor(0.1)),
                                       num_integration_steps=2, hidden_layers=[128, 128]) for _ in range(2)])
        shift_fn, scale_fn = make_shift_fn(nd, hidden_layers=[128, 128]), make_scale_fn(nd, hidden_layers=[128, 128])
        lift = lambda q: tfd.MultivariateNormalDiag(loc=shift_fn(q), scale_diag=scale_fn(q))
        sur = NormalizingFlow(model, flow_bijector=flows, extra_ndims=nd, posterior_lift_distribution=lift)
    else:
        sur.model = model
    lo, hi = rows if rows is not None else (0, r['n'])
    return model, VI(model, TensorDataset(r['X'][lo:hi], r['y'][lo:hi]), sur)


def test_hnf_elbo_step_10m_rows_256_samples(regression_10m, gpu_device):
    """configs[4] with the flow surrogate at full size, through VI.loss / VI._step:
    (i) loss and the gradient of EVERY trainable variable (flow step sizes, energy networks, lift networks) over two
        row shards add up to those of the whole dataset once the data-free part of the loss (log q - log prior - lift
        density) is counted once -- what the all-reduce of partial log-likelihoods under the autograd node relies on;
    (ii) one optimisation step replayed as a HIP graph leaves the variables the eagerly executed step leaves."""
    from bayes_vi_amd.inference.vi.vi import Adam
    r = regression_10m
    n, d, S = r['n'], r['d'], 256
    pu = d + 2
    g = torch.Generator(device=gpu_device); g.manual_seed(123)
    eps = torch.randn(S, 2 * pu, generator=g, device=gpu_device)
    model, vi = _hnf_setup(r, gpu_device)
    sur = vi.surrogate_posterior
    sur.base_noise = lambda m, dim: eps
    params = [p for p in sur.parameters() if p.requires_grad]
    assert len(params) >= 2 * (1 + 2 * 6) + 2 * 6          # 2 flows x (step + two 3-layer MLPs) + two 3-layer lift MLPs

    def loss_and_grads(v):
        for p in params:
            p.grad = None
        l = v.loss(S); l.backward()
        return l.detach().double(), [torch.zeros_like(p, dtype=torch.float64) if p.grad is None else p.grad.double().clone()
                                     for p in params]
    full_l, full_g = loss_and_grads(vi)
    assert bool(torch.isfinite(full_l)) and all(bool(torch.isfinite(t).all()) for t in full_g)
    assert sum(float(t.abs().max()) > 0 for t in full_g) >= len(params) - 2
    cut = 6_100_003
    acc_l, acc_g = torch.zeros((), dtype=torch.float64, device=gpu_device), [torch.zeros_like(t) for t in full_g]
    for rows in ((0, cut), (cut, n)):
        _, vs = _hnf_setup(r, gpu_device, sur=sur, rows=rows)
        l, gs = loss_and_grads(vs)
        acc_l += l
        for a, t in zip(acc_g, gs):
            a += t
        del vs
    # the data-free part, counted twice above: the same loss with the likelihood left out of the target
    p_c = model.flat_param_event_ndims
    prior_only = sur.get_corrected_target_log_prob_fn(lambda sample: vi.posterior.log_prior(sample[:, :p_c], False)[0])
    for p in params:
        p.grad = None
    q, logq = sur.sample_and_log_prob(S)
    l0 = (logq - prior_only(q)).mean(); l0.backward()
    g0 = [torch.zeros_like(t) if p.grad is None else p.grad.double() for p, t in zip(params, full_g)]
    l0_value = l0.detach().double()
    # nothing may keep this autograd graph alive into the capture below: its AccumulateGrad nodes belong to the
    # default stream, and torch then synchronises the capturing stream with it
    del l0, q, logq
    assert abs(float(acc_l - l0_value - full_l)) <= 2 * RTOL * abs(float(full_l))
    worst = 0.0
    for a, z0, f in zip(acc_g, g0, full_g):
        scale = float(f.abs().max())
        if scale == 0.0:
            assert float((a - z0).abs().max()) == 0.0
            continue
        worst = max(worst, float((a - z0 - f).abs().max()) / scale)
    # every variable's gradient is a linear image of d loss / d sample, which adds over shards to 1e-5 of its largest
    # entry; the float32 backward passes through the networks round each of the three runs separately
    assert worst <= 1e-4, worst
    # (ii) eager step against graph replay from the same state, in the order VI.fit uses: a few eager steps first
    # (every kernel and autograd path has run once), then capture of the same step with the same optimizer state
    opt = Adam(learning_rate=1e-3).bind(params)
    for _ in range(3):
        vi._step(opt, params, S)

    def snapshot():
        return ([p.detach().clone() for p in params], [t.clone() for t in opt.m], [t.clone() for t in opt.v],
                opt.t.clone())

    def restore(snap):
        with torch.no_grad():
            for dst, src in zip(params, snap[0]):
                dst.copy_(src)
            for dst, src in zip(opt.m + opt.v, snap[1] + snap[2]):
                dst.copy_(src)
            opt.t.copy_(snap[3])
    state0 = snapshot()
    loss_e = vi._step(opt, params, S).clone()
    eager = [p.detach().clone() for p in params]
    assert any(not torch.equal(a, b) for a, b in zip(eager, state0[0]))
    restore(state0)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = vi._step(opt, params, S)
    restore(state0)
    graph.replay()
    torch.cuda.synchronize()
    assert float(opt.t) == float(state0[3]) + 1.0
    assert float((static_loss - loss_e).abs()) <= 1e-6 * float(loss_e.abs())
    for p, e in zip(params, eager):
        # an Adam step moves every entry by ~lr; the two executions may differ in the last bits of the gradients
        assert float((p.detach() - e).abs().max()) <= 1e-3 * 1e-3 + 1e-6 * float(e.abs().max())
