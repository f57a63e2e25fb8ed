"""GPU parity: the HIP path, called through the C ABI (ctypes), against the fp64
closed-form oracle on the same seeded inputs and against the golden vectors."""
import json
import os

import math

import numpy as np
import pytest
import torch

from oracle import closed_form as cf
from parity_util import assert_grad_close, assert_logp_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def nat(gpu_device):
    from bayes_vi_amd import _native
    _native.lib()
    return _native


@pytest.fixture(scope='module')
def ws(nat, gpu_device):
    return nat.Workspace(gpu_device)


def dev(a, device):
    return torch.tensor(np.asarray(a, np.float32), device=device)


def test_golden_vectors(nat, ws, gpu_device):
    with open(os.path.join(ROOT, 'tests', 'golden', 'hotpath_small.json')) as f:
        cases = json.load(f)['cases']
    assert {c['family'] for c in cases} == {'normal_iid', 'linreg', 'gmm1d', 'gmm_fullcov', 'mvn_tril'}
    for case in cases:
        th = dev(case['theta'], gpu_device)
        if case['family'] == 'gmm_fullcov':
            lp, g = nat.gmm_fullcov_logp_grad(dev(case['Y'], gpu_device), th, case['k'], ws)
            assert_logp_close(lp.cpu().numpy(), case['logp'])
            assert_grad_close(g.cpu().numpy(), case['grad'], term_scale=np.full_like(np.asarray(case['grad']),
                                                                                      3.0 * len(case['Y'])))
            continue
        if case['family'] == 'mvn_tril':
            lp, g = nat.mvn_tril_logp_grad(dev(case['Y'], gpu_device), th, nat.MVN_CHOL_PRECISION_SOLVE, ws)
            assert_logp_close(lp.cpu().numpy(), case['logp'])
            assert_grad_close(g.cpu().numpy(), case['grad'])
            continue
        y = dev(case['y'], gpu_device)
        if case['family'] == 'normal_iid':
            lp, g = nat.normal_iid_logp_grad(y, th, ws)
        elif case['family'] == 'linreg':
            lp, g = nat.linreg_logp_grad(dev(case['X'], gpu_device), y, th, ws)
        else:
            lp, g = nat.gmm1d_logp_grad(y, th, case['k'], ws)
        assert_logp_close(lp.cpu().numpy(), case['logp'])
        assert_grad_close(g.cpu().numpy(), case['grad'])


@pytest.mark.parametrize('c,n', [(1, 1), (2, 20), (5, 100), (3, 1000), (64, 4096), (130, 10007), (17, 300001)])
def test_normal_iid(nat, ws, gpu_device, c, n):
    rng = np.random.default_rng(100 + c + n)
    y = rng.normal(7, 3, n).astype(np.float32)
    th = np.stack([rng.normal(7, 0.5, c), rng.uniform(2, 4, c)], -1).astype(np.float32)
    ref_lp, ref_g = cf.normal_iid(y, th)
    lp, g = nat.normal_iid_logp_grad(dev(y, gpu_device), dev(th, gpu_device), ws)
    r = np.abs(y[None] - th[:, :1]); sc = th[:, 1:2].astype(np.float64)
    assert_logp_close(lp.cpu().numpy(), ref_lp)
    assert_grad_close(g.cpu().numpy(), ref_g, term_scale=np.stack([(r / sc ** 2).sum(-1), (r * r / sc[:, :1] ** 3).sum(-1) + n / sc[:, 0]], -1))
    lp2, g2 = nat.normal_iid_logp_grad(dev(y, gpu_device), dev(th, gpu_device), ws, want_grad=False)
    assert g2 is None
    np.testing.assert_array_equal(lp2.cpu().numpy(), lp.cpu().numpy())


def test_normal_iid_empty(nat, ws, gpu_device):
    th = dev([[1.0, 2.0], [0.5, 1.5]], gpu_device)
    lp, g = nat.normal_iid_logp_grad(torch.empty(0, device=gpu_device), th, ws)
    np.testing.assert_array_equal(lp.cpu().numpy(), [0.0, 0.0])
    np.testing.assert_array_equal(g.cpu().numpy(), np.zeros((2, 2)))


@pytest.mark.parametrize('c', [1, 3, 16, 70])
def test_normal_iid_unaligned_rows_and_short_inputs(nat, ws, gpu_device, c):
    """Row pointers off the 16-byte grid (a view into a larger tensor) and inputs shorter than one
    vector load: head rows, whole 16-byte groups and tail rows must each be counted exactly once."""
    rng = np.random.default_rng(500 + c)
    base = rng.normal(-3, 2, 5000).astype(np.float32)
    base_dev = dev(base, gpu_device)
    th = np.stack([rng.normal(-3, 0.5, c), rng.uniform(1, 3, c)], -1).astype(np.float32)
    for off in (0, 1, 2, 3):
        for n in (1, 2, 3, 4, 5, 7, 8, 9, 1003, 4997 - off):
            y = base[off:off + n]
            ref_lp, ref_g = cf.normal_iid(y, th)
            lp, g = nat.normal_iid_logp_grad(base_dev[off:off + n], dev(th, gpu_device), ws)
            r = np.abs(y[None] - th[:, :1]); sc = th[:, 1:2].astype(np.float64)
            assert_logp_close(lp.cpu().numpy(), ref_lp)
            assert_grad_close(g.cpu().numpy(), ref_g,
                              term_scale=np.stack([(r / sc ** 2).sum(-1), (r * r / sc[:, :1] ** 3).sum(-1) + n / sc[:, 0]], -1))


def _linreg_case(rng, n, d, c, spread=0.05):
    X = rng.normal(size=(n, d)).astype(np.float32)
    w = rng.normal(size=d)
    y = (0.5 + X @ w + rng.normal(size=n)).astype(np.float32)
    th = np.concatenate([0.5 + spread * rng.normal(size=(c, 1)), w + spread * rng.normal(size=(c, d)),
                         rng.uniform(0.8, 1.5, (c, 1))], -1).astype(np.float32)
    return X, y, th


def _linreg_term_scale(X, y, th):
    d = X.shape[1]
    X64, y64, th64 = X.astype(np.float64), y.astype(np.float64), th.astype(np.float64)
    r = np.abs(y64[None] - th64[:, :1] - th64[:, 1:1 + d] @ X64.T)
    s2 = th64[:, -1:] ** 2
    gs = np.concatenate([r.sum(-1, keepdims=True), r @ np.abs(X64),
                         (r * r).sum(-1, keepdims=True) / th64[:, -1:] + len(y) * th64[:, -1:]], -1) / s2
    return gs


@pytest.mark.parametrize('d,c,n', [(1, 2, 50), (1, 1, 1), (2, 5, 100), (3, 16, 777), (5, 33, 4096), (8, 64, 1000),
                                   (8, 100, 20011), (4, 1, 1003), (4, 4, 5001), (8, 3, 2049), (2, 1, 777),
                                   (1, 64, 5000), (2, 70, 3001), (4, 64, 4100), (1, 4, 100003), (12, 4, 500), (16, 16, 1024), (31, 7, 333), (32, 33, 2048),
                                   (33, 5, 100), (48, 64, 4099), (64, 2, 20), (64, 100, 8192), (64, 17, 100003)])
def test_linreg_f32(nat, ws, gpu_device, d, c, n):
    rng = np.random.default_rng(7 * d + c + n)
    X, y, th = _linreg_case(rng, n, d, c)
    ref_lp, ref_g = cf.linreg(X, y, th)
    lp, g = nat.linreg_logp_grad(dev(X, gpu_device), dev(y, gpu_device), dev(th, gpu_device), ws,
                                 precision=nat.PREC_F32)
    assert_logp_close(lp.cpu().numpy(), ref_lp)
    assert_grad_close(g.cpu().numpy(), ref_g, term_scale=_linreg_term_scale(X, y, th))


@pytest.mark.parametrize('d,c,n', [(64, 128, 64), (64, 256, 1000), (64, 300, 4096 + 17), (32, 128, 5000),
                                   (64, 1024, 20000), (32, 513, 777), (64, 32, 300), (64, 130, 200000)])
def test_linreg_bf16_split(nat, ws, gpu_device, d, c, n):
    """BF16X6 (3-piece residual GEMM, 2-piece gradient GEMM) must meet the 1e-5 bar
    like the exact-f32 kernels; AUTO selects it from 128 chains and 16384 rows on."""
    rng = np.random.default_rng(11 * d + c + n)
    X, y, th = _linreg_case(rng, n, d, c, spread=0.05 if n < 100000 else 0.005)
    ref_lp, ref_g = cf.linreg(X, y, th)
    args = (dev(X, gpu_device), dev(y, gpu_device), dev(th, gpu_device))
    lp, g = nat.linreg_logp_grad(*args, ws, precision=nat.PREC_BF16X6)
    assert_logp_close(lp.cpu().numpy(), ref_lp)
    # (this explicitly selected mode keeps the floor any f32 summation has, 3e-7 x the sum of |terms|: one case,
    # 130 chains on 200k rows, needs it; the default kernels are judged without)
    assert_grad_close(g.cpu().numpy(), ref_g, term_scale=_linreg_term_scale(X, y, th), abs_term=3e-7)
    # the packed entry point runs the same arithmetic on a dataset split once up front
    pk = nat.linreg_pack(args[0], args[1])
    lp_p, g_p = nat.linreg_packed_logp_grad(pk, n, d, args[2], ws, precision=nat.PREC_BF16X6)
    np.testing.assert_array_equal(lp_p.cpu().numpy(), lp.cpu().numpy())
    np.testing.assert_array_equal(g_p.cpu().numpy(), g.cpu().numpy())
    lp_a, g_a = nat.linreg_logp_grad(*args, ws)
    lp_f, g_f = nat.linreg_logp_grad(*args, ws, precision=nat.PREC_F32)
    same_as = (lp, g) if (c >= 128 and n >= 16384) else (lp_f, g_f)
    np.testing.assert_array_equal(lp_a.cpu().numpy(), same_as[0].cpu().numpy())
    np.testing.assert_array_equal(g_a.cpu().numpy(), same_as[1].cpu().numpy())


def _f16_eval(nat, ws, gpu_device, X, y, th):
    Xd, yd, thd = dev(X, gpu_device), dev(y, gpu_device), dev(th, gpu_device)
    pk = nat.linreg_pack(Xd, yd, fmt=nat.PACK_F16)
    lp, g = nat.linreg_packed_logp_grad(pk, X.shape[0], X.shape[1], thd, ws, fmt=nat.PACK_F16)
    return lp.cpu().numpy(), g.cpu().numpy()


@pytest.mark.parametrize('d,c,n', [(64, 1, 1), (64, 128, 64), (64, 256, 1000), (64, 300, 4096 + 17), (32, 128, 5000),
                                   (64, 1024, 20000), (32, 513, 777), (64, 32, 300), (64, 130, 200000),
                                   (32, 3, 63), (64, 257, 65), (1, 130, 3000), (5, 128, 20000), (17, 200, 4097),
                                   (33, 129, 1000), (48, 256, 30000), (63, 64, 500), (64, 17, 3000), (64, 64, 20000),
                                   (32, 40, 5000), (17, 33, 4097), (64, 65, 2000)])
def test_linreg_f16_pieces(nat, ws, gpu_device, d, c, n):
    """The f16-piece kernel (X, W and the residuals in 2 pieces each, the remainder of W compensated in closed
    form; power-of-two range scaling) must meet the same 1e-5 bar as the exact-f32 kernels.  At most 64 chains
    run one 16-chain tile per wave, up to 128 two tiles in 4-wave workgroups, more in 8-wave workgroups."""
    rng = np.random.default_rng(7 * d + c + n)
    X, y, th = _linreg_case(rng, n, d, c, spread=0.05 if n < 100000 else 0.005)
    ref_lp, ref_g = cf.linreg(X, y, th)
    lp, g = _f16_eval(nat, ws, gpu_device, X, y, th)
    assert_logp_close(lp, ref_lp)
    assert_grad_close(g, ref_g, term_scale=_linreg_term_scale(X, y, th))


# The three tests below compare the f16-plane kernel with the exact-f32 kernel's own error against the float64 oracle:
# any float32 evaluation rounds the fitted values to 1e-7 of their size, which near the mode -- where the gradient is
# a small difference of large sums -- is a floor above the 1e-5 bar ("f32 floor": the reference's float32 TFP graph
# sits on it as well).  A ratio alone would let both kernels regress together, so every case also has an absolute
# ceiling: the error measured when the test was written (MI355X, the seeds below) with a factor 2 of headroom.
F32_FLOOR_CEILINGS = {          # case: (f16-plane kernel, exact-f32 kernel); measured values in the comments
    'near mode, 128 chains': (1.1e-6, 1.1e-5),            # 5.2e-7, 5.4e-6
    'near mode, 48 chains': (6e-7, 1.2e-5),               # 2.6e-7, 5.6e-6
    'at the mode, small weights': (3.2e-6, 2.8e-6),       # 1.6e-6, 1.4e-6
    'signal / noise, sigma 0.1': (1.4e-5, 4.5e-5),        # 7.0e-6, 2.2e-5
    'signal / noise, sigma 0.01': (2.7e-4, 3.6e-4),       # 1.3e-4, 1.8e-4
    'signal / noise, sigma 0.001': (2.2e-3, 3.5e-3),      # 1.1e-3, 1.7e-3
}


def _check_f32_floor(case, e16, e32, ratio, slack=0.0):
    print('f32 floor [%s]: f16-plane kernel %.3e, exact-f32 kernel %.3e (of the chain\'s largest gradient entry)'
          % (case, e16, e32))
    assert e16 <= ratio * e32 + slack, (case, e16, e32)
    ceiling = F32_FLOOR_CEILINGS.get(case)
    assert ceiling is not None, 'no recorded ceiling for %s: measured f16 %.3e, f32 %.3e' % (case, e16, e32)
    assert e16 <= ceiling[0] and e32 <= ceiling[1], (case, e16, e32, ceiling)


@pytest.mark.parametrize('c', [128, 48])
def test_linreg_f16_pieces_near_mode(nat, ws, gpu_device, c):
    """Chains packed tightly around the mode: the gradient is a small difference of large sums and an
    error in W common to all rows would grow like n (the reason the remainder of W after two pieces is
    compensated with X^T X).  The error must stay within a small multiple of the exact-f32 kernel's own."""
    rng = np.random.default_rng(99)
    n, d = 400000, 64
    X, y, th = _linreg_case(rng, n, d, c, spread=0.001)
    ref_lp, ref_g = cf.linreg(X, y, th)
    lp, g = _f16_eval(nat, ws, gpu_device, X, y, th)
    lp32, g32 = nat.linreg_logp_grad(dev(X, gpu_device), dev(y, gpu_device), dev(th, gpu_device), ws,
                                     precision=nat.PREC_F32)
    scale = np.abs(ref_g).max(-1, keepdims=True)
    e16 = (np.abs(g - ref_g) / scale).max()
    e32 = (np.abs(g32.cpu().numpy() - ref_g) / scale).max()
    assert_logp_close(lp, ref_lp)
    _check_f32_floor('near mode, %d chains' % c, e16, max(e32, 2.5e-6), 4.0)


def test_linreg_f16_pieces_at_the_mode_small_weights(nat, ws, gpu_device):
    """Chains within one posterior standard deviation of the mode, weights spanning three decades: the
    matrix instruction cuts small addends off toward zero, which for full-mantissa pieces of the small weights
    was a loss with a sign (gradient error on exactly those features, the same for every chain, 10x the
    exact-f32 kernel's).  With the first piece of W on a grid relative to the chain's largest weight the error
    must stay within a small multiple of the exact-f32 kernel's own."""
    rng = np.random.default_rng(17)
    n, d, c = 300000, 64, 48
    X = rng.normal(size=(n, d)).astype(np.float32)
    w = rng.normal(size=d) * 10.0 ** rng.uniform(-3, 0, d)
    y = (0.5 + X.astype(np.float64) @ w + rng.normal(size=n)).astype(np.float32)
    A = np.concatenate([np.ones((n, 1)), X.astype(np.float64)], 1)
    ols = np.linalg.lstsq(A, y.astype(np.float64), rcond=None)[0]
    width = 1.0 / math.sqrt(n)
    th = np.concatenate([ols[None, :1] + width * rng.normal(size=(c, 1)), ols[None, 1:] + width * rng.normal(size=(c, d)),
                         np.ones((c, 1))], -1).astype(np.float32)
    ref_lp, ref_g = cf.linreg(X, y, th)
    lp, g = _f16_eval(nat, ws, gpu_device, X, y, th)
    lp32, g32 = nat.linreg_logp_grad(dev(X, gpu_device), dev(y, gpu_device), dev(th, gpu_device), ws,
                                     precision=nat.PREC_F32)
    scale = np.abs(ref_g[:, 1:1 + d]).max(-1, keepdims=True)
    e16 = (np.abs(g - ref_g)[:, 1:1 + d] / scale).max()
    e32 = (np.abs(g32.cpu().numpy() - ref_g)[:, 1:1 + d] / scale).max()
    assert_logp_close(lp, ref_lp)
    _check_f32_floor('at the mode, small weights', e16, e32, 3.0)


@pytest.mark.parametrize('sigma', [0.1, 0.01, 0.001])
def test_linreg_f16_pieces_high_signal_to_noise(nat, ws, gpu_device, sigma):
    """Residuals far smaller than the fitted values (signal / noise up to ~5000), chains at the posterior's own
    width around the mode.  The kernel streams one f16 plane of X against one f16 piece of W and adds what both
    leave out in closed form, so its own residuals carry an error of ~2^-12 of the fitted values -- here up to
    the size of the true residuals -- which the closed form has to remove exactly: the log-density must stay at
    f32 accuracy and the gradient within a small multiple of the exact-f32 kernel's error (which grows with
    signal / noise in the same way)."""
    rng = np.random.default_rng(17)
    n, d, c = 200000, 64, 40
    X = rng.normal(size=(n, d)).astype(np.float32)
    w = rng.normal(size=d)
    y = (0.5 + X.astype(np.float64) @ w + sigma * rng.normal(size=n)).astype(np.float32)
    A = np.concatenate([np.ones((n, 1)), X.astype(np.float64)], 1)
    ols = np.linalg.lstsq(A, y.astype(np.float64), rcond=None)[0]
    width = sigma / math.sqrt(n)
    th = np.concatenate([ols[None, :1] + width * rng.normal(size=(c, 1)), ols[None, 1:] + width * rng.normal(size=(c, d)),
                         sigma * rng.uniform(0.9, 1.1, (c, 1))], -1).astype(np.float32)
    ref_lp, ref_g = cf.linreg(X, y, th)
    lp, g = _f16_eval(nat, ws, gpu_device, X, y, th)
    lp32, g32 = nat.linreg_logp_grad(dev(X, gpu_device), dev(y, gpu_device), dev(th, gpu_device), ws,
                                     precision=nat.PREC_F32)
    scale = np.abs(ref_g[:, 1:1 + d]).max(-1, keepdims=True)
    e16 = (np.abs(g - ref_g)[:, 1:1 + d] / scale).max()
    e32 = (np.abs(g32.cpu().numpy() - ref_g)[:, 1:1 + d] / scale).max()
    _check_f32_floor('signal / noise, sigma %g' % sigma, e16, e32, 2.0, slack=1e-6)
    # the log-density: inside the 1e-5 bar with room to spare, and at the level of the exact-f32 kernel's own error
    # (at signal / noise ~ 8000 both are set by the f32 rounding of the fitted values: ~1.5e-6 for that kernel)
    rel = (np.abs(lp - ref_lp) / np.abs(ref_lp)).max()
    rel32 = (np.abs(lp32.cpu().numpy() - ref_lp) / np.abs(ref_lp)).max()
    assert rel <= 5e-6 and rel <= 2.0 * rel32 + 1e-6, (rel, rel32)
    # d logp / d scale = -n / s + S2 / s^3 is a difference of two equal terms at the mode, so it shows the error of
    # S2 itself; any f32 evaluation rounds the fitted values to ~1e-7 of their size, which is ~1e-3 of a residual
    # at signal / noise 5000 and averages down with sqrt(n): the exact-f32 kernel sets the scale here too
    gs = (np.abs(g[:, d + 1] - ref_g[:, d + 1]) / (n / th[:, d + 1])).max()
    gs32 = (np.abs(g32.cpu().numpy()[:, d + 1] - ref_g[:, d + 1]) / (n / th[:, d + 1])).max()
    assert gs <= 2.0 * gs32 + 1e-6, (gs, gs32)


def test_linreg_f16_pieces_column_ranges(nat, ws, gpu_device):
    """Columns six decades apart in scale, off-centre, one heavy-tailed, one constant, one all zero;
    targets in the hundreds: the per-column / per-chain / residual scales must absorb all of it."""
    rng = np.random.default_rng(123)
    n, d, c = 30000, 64, 160
    colscale = 10.0 ** rng.uniform(-3, 3, d)
    X = rng.normal(size=(n, d)) + rng.normal(size=d) * 2.0
    X[:, 5] = np.exp(1.5 * rng.normal(size=n))        # max / median ~ 1e3
    X[:, 6] = 1.0
    X = X * colscale
    X[:, 7] = 0.0
    X = X.astype(np.float32)
    w = rng.normal(size=d) / colscale
    y = (300.0 + X.astype(np.float64) @ w + 3.0 * rng.normal(size=n)).astype(np.float32)
    th = np.concatenate([300.0 + 0.3 * rng.normal(size=(c, 1)), w * (1 + 0.01 * rng.normal(size=(c, d))),
                         rng.uniform(2.0, 4.0, (c, 1))], -1).astype(np.float32)
    ref_lp, ref_g = cf.linreg(X, y, th)
    lp, g = _f16_eval(nat, ws, gpu_device, X, y, th)
    assert_logp_close(lp, ref_lp)
    # gradient entries differ by the column scales: judge every entry against its own term scale
    ts = _linreg_term_scale(X, y, th)
    err = np.abs(g - ref_g)
    assert (err <= 1e-5 * np.abs(ref_g) + 3e-7 * ts + 1e-30).all(), float((err / (np.abs(ref_g) + ts * 3e-2)).max())


@pytest.mark.parametrize('c', [256, 130])
def test_linreg_f16_block_scale_and_per_chain_scale_agree(nat, ws, gpu_device, c):
    """Many-chain blocks whose chains have natural scales within 2^8 of each other work with one scale and no
    intercept in the streaming pass (targets enter through the accumulator, the intercept in closed form); a block
    with one chain a thousand times further out keeps one scale and one intercept per chain.  The other chains
    must come out the same either way, and both ways within the bar of the float64 closed form."""
    rng = np.random.default_rng(41)
    n, d = 30000, 64
    X, y, th = _linreg_case(rng, n, d, c, spread=0.02)
    far = th.copy()
    far[7, 1:1 + d] *= 1000.0            # scale 2^10 below the others': the block takes the per-chain route
    ref_lp, ref_g = cf.linreg(X, y, far)
    lp_u, g_u = _f16_eval(nat, ws, gpu_device, X, y, th)
    lp_f, g_f = _f16_eval(nat, ws, gpu_device, X, y, far)
    assert_logp_close(lp_f, ref_lp)
    assert_grad_close(g_f, ref_g, term_scale=_linreg_term_scale(X, y, far))
    near = np.ones(c, bool); near[7] = False
    assert (np.abs(lp_u[near] - lp_f[near]) <= 2e-6 * np.abs(lp_f[near])).all()
    assert_logp_close(lp_u[near], ref_lp[near])
    assert (np.abs(g_u[near] - g_f[near]).max(-1) <= 1e-5 * np.abs(ref_g[near]).max(-1)).all()
    ref_lp_u, ref_g_u = cf.linreg(X, y, th)
    assert_logp_close(lp_u, ref_lp_u)
    assert_grad_close(g_u, ref_g_u, term_scale=_linreg_term_scale(X, y, th))


@pytest.mark.parametrize('c', [4, 160])
def test_linreg_f16_columns_far_off_centre(nat, ws, gpu_device, c):
    """Columns whose mean is 30 - 1000 times their spread, chains near the mode and chains scattered around it:
    the packed plane holds the centred columns and the kernels work with the intercept at the centroid, so
    neither the f16 range nor the closed-form intercept loses anything to the offsets."""
    rng = np.random.default_rng(77)
    n, d = 60000, 64
    offs = rng.choice([-1.0, 1.0], d) * 10.0 ** rng.uniform(1.5, 3.0, d)
    X = (rng.normal(size=(n, d)) + offs).astype(np.float32)
    w = rng.normal(size=d)
    y = (2.0 + X.astype(np.float64) @ w + rng.normal(size=n)).astype(np.float32)
    A = np.concatenate([np.ones((n, 1)), X.astype(np.float64)], 1)
    ols = np.linalg.lstsq(A, y.astype(np.float64), rcond=None)[0]
    width = np.where(np.arange(c) % 2 == 0, 1.0 / math.sqrt(n), 0.05)[:, None]
    # scattered chains keep their fit at the centroid: the intercept follows the weights
    dw = width * rng.normal(size=(c, d))
    b0 = ols[0] - dw @ X.astype(np.float64).mean(0) + width[:, 0] * rng.normal(size=c)
    th = np.concatenate([b0[:, None], ols[None, 1:] + dw, rng.uniform(0.9, 1.2, (c, 1))], -1).astype(np.float32)
    ref_lp, ref_g = cf.linreg(X, y, th)
    lp, g = _f16_eval(nat, ws, gpu_device, X, y, th)
    assert_logp_close(lp, ref_lp)
    assert_grad_close(g, ref_g, term_scale=_linreg_term_scale(X, y, th))


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_linreg_f16_random_shapes(nat, ws, gpu_device, seed):
    """Random feature counts (9 - 64), chain counts around every block boundary (1 ... 300), ragged row counts,
    columns off-centre by up to 300 times their spread and six decades apart in scale, chains from 1e-3 to 0.3 of
    the weights around the fit, now and then one chain thousands of times further out (its block then keeps one
    scale and intercept per chain): value + gradient and value only against the float64 closed form."""
    rng = np.random.default_rng(seed)
    for _ in range(12):
        d = int(rng.integers(9, 65)); c = int(rng.choice([1, 3, 16, 17, 33, 64, 65, 100, 129, 256, 257, 300]))
        n = int(rng.choice([50, 127, 128, 129, 1000, 4097, 20000, 70001]))
        off = rng.normal(size=d) * 10.0 ** rng.uniform(-1, 2.5)
        X = (rng.normal(size=(n, d)) * 10.0 ** rng.uniform(-2, 2, d) + off).astype(np.float32)
        w = rng.normal(size=d) / np.maximum(np.abs(X).max(0), 1e-3)
        y = (rng.normal() * 5 + X.astype(np.float64) @ w + rng.normal(size=n) * 10.0 ** rng.uniform(-1, 0.5)).astype(np.float32)
        spread = 10.0 ** rng.uniform(-3, -0.5)
        th = np.concatenate([y.mean() - off @ w + spread * rng.normal(size=(c, 1)),
                             w * (1 + spread * rng.normal(size=(c, d))), rng.uniform(0.5, 2.0, (c, 1))], -1).astype(np.float32)
        if rng.random() < 0.3 and c > 2:
            th[rng.integers(c), 1:1 + d] *= 3000.0
        ref_lp, ref_g = cf.linreg(X, y, th)
        Xd, yd, thd = dev(X, gpu_device), dev(y, gpu_device), dev(th, gpu_device)
        pk = nat.linreg_pack(Xd, yd, fmt=nat.PACK_F16)
        lp, g = nat.linreg_packed_logp_grad(pk, n, d, thd, ws, fmt=nat.PACK_F16)
        lpv, _ = nat.linreg_packed_logp_grad(pk, n, d, thd, ws, fmt=nat.PACK_F16, want_grad=False)
        assert_logp_close(lp.cpu().numpy(), ref_lp)
        assert_logp_close(lpv.cpu().numpy(), ref_lp)
        assert_grad_close(g.cpu().numpy(), ref_g)


def test_linreg_f16_pieces_edge_behaviour(nat, ws, gpu_device):
    rng = np.random.default_rng(3)
    X, y, th = _linreg_case(rng, 5000, 64, 130)
    th[3, 10] = np.nan
    th[64, 0] = np.inf
    th[100, -1] = -1.0
    Xd, yd, thd = dev(X, gpu_device), dev(y, gpu_device), dev(th, gpu_device)
    pk = nat.linreg_pack(Xd, yd, fmt=nat.PACK_F16)
    lp, g = nat.linreg_packed_logp_grad(pk, 5000, 64, thd, ws, fmt=nat.PACK_F16)
    lp2, g2 = nat.linreg_packed_logp_grad(pk, 5000, 64, thd, ws, fmt=nat.PACK_F16)
    assert torch.equal(lp.isnan(), lp2.isnan()) and torch.equal(lp.nan_to_num(), lp2.nan_to_num())
    assert torch.equal(g.nan_to_num(), g2.nan_to_num())
    lp = lp.cpu().numpy()
    bad = np.zeros(130, bool); bad[[3, 64, 100]] = True
    assert not np.isfinite(lp[bad]).any() and np.isfinite(lp[~bad]).all()
    assert np.isfinite(g.cpu().numpy()[~bad]).all()
    ref_lp, ref_g = cf.linreg(X, y, th[~bad])
    assert_logp_close(lp[~bad], ref_lp)
    # no chains beyond the valid ones are touched, no rows beyond n are read: n = 0 gives the empty sum
    pk0 = nat.linreg_pack(Xd[:0], yd[:0], fmt=nat.PACK_F16)
    lp0, g0 = nat.linreg_packed_logp_grad(pk0, 0, 64, thd[:2], ws, fmt=nat.PACK_F16)
    assert (lp0 == 0).all() and (g0 == 0).all()
    with pytest.raises(nat.BviError):     # this format has one arithmetic; an explicit precision is refused
        nat.linreg_packed_logp_grad(pk, 5000, 64, thd, ws, fmt=nat.PACK_F16, precision=nat.PREC_F32)
    with pytest.raises(nat.BviError):     # more than 64 features: not covered by the packed formats
        nat.linreg_pack(torch.zeros(10, 65, device=gpu_device), torch.zeros(10, device=gpu_device), fmt=nat.PACK_F16)
    with pytest.raises(nat.BviError):     # the bf16 format takes 32 or 64 features only
        nat.linreg_pack(Xd[:, :16].contiguous(), yd, fmt=nat.PACK_BF16X3)


def test_linear_regression_auto_runs_f16_pieces(nat, ws, gpu_device):
    """The host mirror packs once and routes many-chain evaluations on a large shard to the
    f16-piece kernel; few chains stay on the exact-f32 kernel."""
    from bayes_vi_amd.fused_target import LinearRegression
    rng = np.random.default_rng(8)
    X, y, th = _linreg_case(rng, 20000, 64, 256)
    Xd, yd, thd = dev(X, gpu_device), dev(y, gpu_device), dev(th, gpu_device)
    lik = LinearRegression(Xd, yd)
    assert lik.matrix_core_format(256) == nat.PACK_F16 and lik.matrix_core_format(17) == nat.PACK_F16 \
        and lik.matrix_core_format(16) is None
    lik.n = 400_000              # from 400k rows on the f16 plane serves any number of chains
    assert lik.matrix_core_format(16) == nat.PACK_F16 and lik.matrix_core_format(1) == nat.PACK_F16
    lik.n = 20000
    lp, g = lik.logp_grad(thd)
    pk = nat.linreg_pack(Xd, yd, fmt=nat.PACK_F16)
    lp_p, g_p = nat.linreg_packed_logp_grad(pk, 20000, 64, thd, ws, fmt=nat.PACK_F16)
    assert torch.equal(lp, lp_p) and torch.equal(g, g_p)
    lp_s, g_s = lik.logp_grad(thd[:16].contiguous())
    lp_f, g_f = nat.linreg_logp_grad(Xd, yd, thd[:16].contiguous(), ws, precision=nat.PREC_F32)
    assert torch.equal(lp_s, lp_f) and torch.equal(g_s, g_f)


@pytest.mark.parametrize('d,c,n', [(64, 128, 5000), (32, 256, 20000)])
def test_linreg_bf16x3_reduced_precision(nat, ws, gpu_device, d, c, n):
    """BF16X3 (2-piece split in both GEMMs) is the documented reduced-precision mode:
    ~2^-18 per product leaves the residuals good to ~1e-5 absolute, so logp still meets
    1e-5 but the gradient, a cancelling sum, is only good to ~1e-3 of its largest entry.
    It is never selected by AUTO."""
    rng = np.random.default_rng(11 * d + c + n)
    X, y, th = _linreg_case(rng, n, d, c)
    ref_lp, ref_g = cf.linreg(X, y, th)
    lp, g = nat.linreg_logp_grad(dev(X, gpu_device), dev(y, gpu_device), dev(th, gpu_device), ws,
                                 precision=nat.PREC_BF16X3)
    assert_logp_close(lp.cpu().numpy(), ref_lp)
    assert_grad_close(g.cpu().numpy(), ref_g, rtol=2e-3)


def test_linreg_deterministic(nat, ws, gpu_device):
    rng = np.random.default_rng(5)
    X, y, th = _linreg_case(rng, 30000, 64, 256)
    args = (dev(X, gpu_device), dev(y, gpu_device), dev(th, gpu_device))
    a = nat.linreg_logp_grad(*args, ws)
    b = nat.linreg_logp_grad(*args, ws)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_linreg_pack_is_deterministic_and_its_sums_match_float64(nat, ws, gpu_device):
    """The pack step (column statistics, f16 plane, and the pack-time sums X^T X, A = X^T X - XH^T XH, Bt = XH^T XR,
    XR^T y, XR^T 1 behind the header, taken on the matrix cores as per-tile f32 sums added in double in a fixed order)
    writes the same bytes on every run, for row counts that do and do not fill the last tile and for feature
    counts below the plane's width; X^T X of the centred columns agrees with a float64 evaluation."""
    rng = np.random.default_rng(77)
    for n, d in ((70001, 64), (4096, 32), (12345, 50), (300, 20)):
        X = (rng.normal(size=(n, d)) * 10.0 ** rng.uniform(-2, 2, d) + rng.normal(size=d)).astype(np.float32)
        y = (X @ rng.normal(size=d) + rng.normal(size=n)).astype(np.float32)
        Xd, yd = dev(X, gpu_device), dev(y, gpu_device)
        a = nat.linreg_pack(Xd, yd, fmt=nat.PACK_F16)
        b = nat.linreg_pack(Xd, yd, fmt=nat.PACK_F16)
        # (bytes 1040 .. 1279, between the header's fields and the sums, are never written)
        assert torch.equal(a[:1040], b[:1040]) and torch.equal(a[1280:], b[1280:]), (n, d)
        hdr = a[:1280 + 64 * 64 * 8].cpu().numpy()
        shift = hdr[64 * 4 + 64 * 8 + 8 + 4 + 4:][:64 * 4].view(np.float32)[:d].astype(np.float64)
        gram = hdr[1280:].view(np.float64).reshape(64, 64)
        xc = X.astype(np.float64) - shift
        ref = xc.T @ xc
        scale = np.sqrt(np.outer(np.diag(ref), np.diag(ref)))
        # a tile sum of squares comes out of the f32 matrix instruction about one float ulp low (its accumulator is cut
        # off, not rounded), the same way in every tile: ~1e-7 of X^T X's diagonal, which only ever multiplies the
        # 2^-13-sized remainder of the weights
        assert (np.abs(gram[:d, :d] - ref) / scale).max() <= 2.5e-7, (n, d, (np.abs(gram[:d, :d] - ref) / scale).max())
        assert not gram[d:].any() and not gram[:, d:].any()


def test_linreg_errors(nat, ws, gpu_device):
    X = torch.zeros(10, 65, device=gpu_device); y = torch.zeros(10, device=gpu_device)
    th = torch.ones(2, 67, device=gpu_device)
    with pytest.raises(nat.BviError):
        nat.linreg_logp_grad(X, y, th, ws)
    X = torch.zeros(10, 16, device=gpu_device); th = torch.ones(2, 18, device=gpu_device)
    with pytest.raises(nat.BviError):
        nat.linreg_logp_grad(X, y, th, ws, precision=nat.PREC_BF16X6)
    with pytest.raises(nat.BviError):
        nat.linreg_logp_grad(X.cpu(), y, th, ws)


@pytest.mark.parametrize('d', [1, 2, 4, 8])
def test_linreg_small_d_unaligned_rows(nat, ws, gpu_device, d):
    """Few features, few chains: the row-streaming kernel needs 16-byte aligned X and y; a view that starts
    off that grid (a shard cut at an odd row) takes the guarded kernel and gives the same sums."""
    rng = np.random.default_rng(40 + d)
    n = 3001
    X, y, th = _linreg_case(rng, n + 3, d, 3)
    Xd, yd, thd = dev(X, gpu_device), dev(y, gpu_device), dev(th, gpu_device)
    for off in (0, 1, 2, 3):
        Xs, ys = X[off:off + n], y[off:off + n]
        ref_lp, ref_g = cf.linreg(Xs, ys, th)
        lp, g = nat.linreg_logp_grad(Xd[off:off + n], yd[off:off + n], thd, ws, precision=nat.PREC_F32)
        assert_logp_close(lp.cpu().numpy(), ref_lp)
        assert_grad_close(g.cpu().numpy(), ref_g, term_scale=_linreg_term_scale(Xs, ys, th))


def test_streaming_kernels_accept_unaligned_data(nat, ws, gpu_device):
    """A dataset that starts 4 bytes into an allocation (a row shard cut at an odd row): the
    16-byte loads are replaced by guarded scalar ones, the result is unchanged."""
    rng = np.random.default_rng(77)
    n = 5003
    y = rng.normal(2, 3, n + 1).astype(np.float32)
    yd = dev(y, gpu_device)[1:]
    assert yd.data_ptr() % 16 != 0
    th = np.stack([rng.normal(2, .1, 5), rng.uniform(2, 4, 5)], -1).astype(np.float32)
    lp, g = nat.normal_iid_logp_grad(yd, dev(th, gpu_device), ws)
    rlp, rg = cf.normal_iid(y[1:], th)
    assert_logp_close(lp.cpu().numpy(), rlp); assert_grad_close(g.cpu().numpy(), rg)
    k = 3
    thm = np.concatenate([rng.dirichlet(np.ones(k) * 5, 4), np.array([-1., 2., 5.]) + rng.normal(0, .3, (4, k)),
                          rng.uniform(2, 4, (4, k))], -1).astype(np.float32)
    lp, g = nat.gmm1d_logp_grad(yd, dev(thm, gpu_device), k, ws)
    rlp, rg = cf.gmm1d(y[1:], thm, k)
    assert_logp_close(lp.cpu().numpy(), rlp); assert_grad_close(g.cpu().numpy(), rg)


def test_nonfinite_parameters_propagate(nat, ws, gpu_device):
    y = dev(np.linspace(-1, 1, 100), gpu_device)
    th = dev([[0.0, -1.0], [float('nan'), 1.0], [0.0, 1.0]], gpu_device)
    lp, g = nat.normal_iid_logp_grad(y, th, ws)
    lp = lp.cpu().numpy()
    assert not np.isfinite(lp[0]) and not np.isfinite(lp[1]) and np.isfinite(lp[2])


@pytest.mark.parametrize('k,c', [(3, 1), (8, 4), (16, 3), (5, 9)])
def test_gmm1d_unaligned_rows_and_short_inputs(nat, ws, gpu_device, k, c):
    """Views that start off the 16-byte grid and inputs shorter than one vector load (the row-streaming
    kernel of 1 - 4 chains splits the rows into head / 16-byte groups / tail)."""
    rng = np.random.default_rng(900 + k + c)
    locs = np.linspace(-10, 10, k)
    base = rng.normal(locs[rng.integers(0, k, 3000)], 1.0).astype(np.float32)
    base_dev = dev(base, gpu_device)
    th = np.concatenate([rng.dirichlet(np.ones(k) * 5, c), locs + rng.normal(0, .3, (c, k)),
                         rng.uniform(0.8, 1.3, (c, k))], -1).astype(np.float32)
    for off in (0, 1, 2, 3):
        for n in (1, 3, 4, 6, 9, 1001, 2997 - off):
            y = base[off:off + n]
            ref_lp, ref_g = cf.gmm1d(y, th, k)
            lp, g = nat.gmm1d_logp_grad(base_dev[off:off + n], dev(th, gpu_device), k, ws)
            ts = np.concatenate([n / th[:, :k], n / th[:, 2 * k:] * np.ones((c, k)), 3 * n / th[:, 2 * k:]], -1) * 0.3
            assert_logp_close(lp.cpu().numpy(), ref_lp)
            assert_grad_close(g.cpu().numpy(), ref_g, term_scale=ts)


@pytest.mark.parametrize('k,c,n', [(1, 1, 10), (2, 4, 100), (3, 5, 999), (5, 2, 64), (8, 4, 4096), (8, 64, 3000),
                                   (8, 70, 20001), (16, 3, 500), (4, 16, 100000)])
def test_gmm1d(nat, ws, gpu_device, k, c, n):
    rng = np.random.default_rng(13 * k + c + n)
    locs = np.linspace(-20, 20, k) if k > 1 else np.array([1.0])
    scs = 1.0 + (np.arange(k) % 3)
    comp = rng.integers(0, k, n)
    y = rng.normal(locs[comp], scs[comp]).astype(np.float32)
    th = np.concatenate([rng.dirichlet(np.ones(k) * 5, c) * rng.uniform(0.5, 2.0, (c, 1)),
                         locs + rng.normal(0, .5, (c, k)), scs * rng.uniform(0.8, 1.2, (c, k))], -1).astype(np.float32)
    ref_lp, ref_g = cf.gmm1d(y, th, k)
    lp, g = nat.gmm1d_logp_grad(dev(y, gpu_device), dev(th, gpu_device), k, ws)
    # scale of the summed terms: every row contributes O(1/s), O(1/w) sized terms
    ts = np.concatenate([n / th[:, :k], n / th[:, 2 * k:] * np.ones((c, k)), 3 * n / th[:, 2 * k:]], -1) * 0.3
    assert_logp_close(lp.cpu().numpy(), ref_lp)
    assert_grad_close(g.cpu().numpy(), ref_g, term_scale=ts)
    # the value-only instantiations (what random-walk Metropolis calls) return the same log-densities
    lpv, none = nat.gmm1d_logp_grad(dev(y, gpu_device), dev(th, gpu_device), k, ws, want_grad=False)
    assert none is None and torch.equal(lp, lpv)


@pytest.mark.parametrize('d,k,c,n', [(2, 4, 3, 100), (4, 3, 2, 25), (8, 5, 4, 1000), (32, 16, 2, 500), (32, 16, 5, 4099),
                                     (16, 8, 7, 2000), (5, 1, 2, 64), (3, 2, 1, 17), (32, 12, 3, 700), (16, 9, 2, 333)])
def test_gmm_fullcov(nat, ws, gpu_device, d, k, c, n):
    rng = np.random.default_rng(17 * d + k + c + n)
    true_mu = rng.integers(-20, 21, (k, d)).astype(np.float64)
    comp = rng.integers(0, k, n)
    Y = (true_mu[comp] + rng.normal(size=(n, d)) * 1.5).astype(np.float32)
    w = (rng.dirichlet(np.ones(k) * 5, c) * rng.uniform(0.5, 2.0, (c, 1))).astype(np.float32)
    mu = (true_mu[None] + rng.normal(0, 0.3, (c, k, d))).astype(np.float32)
    L = np.tril(rng.normal(size=(c, k, d, d)) * 0.2)
    for i in range(d):
        L[..., i, i] = rng.uniform(1.0, 2.0, (c, k))
    L = L.astype(np.float32)
    ref_lp, (gw, gmu, gL) = cf.gmm_fullcov(Y, w, mu, L)
    theta = np.concatenate([w, mu.reshape(c, -1), L.reshape(c, -1)], -1)
    ref_g = np.concatenate([gw, gmu.reshape(c, -1), gL.reshape(c, -1)], -1)
    lp, g = nat.gmm_fullcov_logp_grad(dev(Y, gpu_device), dev(theta, gpu_device), k, ws)
    assert_logp_close(lp.cpu().numpy(), ref_lp)
    # every row contributes O(1) sized terms to each entry
    assert_grad_close(g.cpu().numpy(), ref_g, term_scale=np.full_like(ref_g, 3.0 * n))
    lp2, g2 = nat.gmm_fullcov_logp_grad(dev(Y, gpu_device), dev(theta, gpu_device), k, ws, want_grad=False)
    assert g2 is None
    np.testing.assert_array_equal(lp2.cpu().numpy(), lp.cpu().numpy())


@pytest.mark.parametrize('n,d,k', [(20000, 32, 16), (777, 5, 3), (40, 8, 2), (9000, 16, 1)])
def test_gmm_fullcov_packed_equals_unpacked(nat, ws, gpu_device, n, d, k):
    """bvi_gmm_fullcov_pack + bvi_gmm_fullcov_packed_logp_grad (the rows' own statistics and f16 pieces taken once)
    against bvi_gmm_fullcov_logp_grad (taken per call): the same bits, with and without gradient, on rows stored
    cluster by cluster and as drawn; and the same for the single multivariate normal."""
    rng = np.random.default_rng(5 * n + d + k)
    c = 5
    true_mu = rng.integers(-12, 13, (k, d)).astype(np.float64)
    Y = (true_mu[rng.integers(0, k, n)] + rng.normal(size=(n, d))).astype(np.float32)
    w = rng.dirichlet(np.ones(k) * 5, c).astype(np.float32)
    mu = (true_mu[None] + rng.normal(0, 0.2, (c, k, d))).astype(np.float32)
    L = np.tril(rng.normal(size=(c, k, d, d)) * 0.05)
    for i in range(d):
        L[..., i, i] = rng.uniform(0.9, 1.3, (c, k))
    theta = dev(np.concatenate([w, mu.reshape(c, -1), L.astype(np.float32).reshape(c, -1)], -1), gpu_device)
    for rows in (dev(Y, gpu_device), nat.cluster_rows(dev(Y, gpu_device), 4 * k)):
        packed = nat.gmm_fullcov_pack(rows)
        lp, g = nat.gmm_fullcov_logp_grad(rows, theta, k, ws)
        lp2, g2 = nat.gmm_fullcov_logp_grad(rows, theta, k, ws, packed=packed)
        assert torch.equal(lp, lp2) and torch.equal(g, g2)
        lpv, _ = nat.gmm_fullcov_logp_grad(rows, theta, k, ws, want_grad=False)
        lpv2, _ = nat.gmm_fullcov_logp_grad(rows, theta, k, ws, want_grad=False, packed=packed)
        assert torch.equal(lpv, lpv2)
        if k == 1:
            th1 = dev(np.concatenate([mu[:, 0], L[:, 0].astype(np.float32).reshape(c, -1)], -1), gpu_device)
            a = nat.mvn_tril_logp_grad(rows, th1, nat.MVN_SCALE_TRIL, ws)
            b = nat.mvn_tril_logp_grad(rows, th1, nat.MVN_SCALE_TRIL, ws, packed=packed)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize('n,d,m', [(50000, 32, 64), (12345, 5, 16), (4097, 1, 128), (100, 7, 128), (1, 3, 16),
                                   (300001, 16, 40)])
def test_cluster_rows_is_a_stable_reproducible_grouping(nat, gpu_device, n, d, m):
    """bvi_cluster_rows: the output holds every row exactly once (order is a permutation and Y_out = Y[order]), rows of
    one group keep their original order (stable), two runs agree bit for bit, a row with a non-finite entry is kept,
    and on well-separated clusters nearly every pair of neighbouring rows comes from the same cluster."""
    rng = np.random.default_rng(n + 7 * d + m)
    kc = 6
    centre = rng.integers(-30, 31, (kc, d)).astype(np.float32) * 3.0
    comp = rng.integers(0, kc, n)
    Y = centre[comp] + rng.normal(size=(n, d)).astype(np.float32)
    if n > 50:
        Y[17, 0] = np.nan
        Y[41, d - 1] = np.inf
    Yd = dev(Y, gpu_device)
    out, order = nat.cluster_rows(Yd, m, want_order=True)
    out2, order2 = nat.cluster_rows(Yd, m, want_order=True)
    assert torch.equal(order, order2) and torch.equal(out.view(torch.int32), out2.view(torch.int32))
    assert torch.equal(torch.sort(order).values, torch.arange(n, device=gpu_device))
    assert torch.equal(out.view(torch.int32), Yd[order].view(torch.int32))
    o = order.cpu().numpy()
    # groups are runs of increasing original index: at most m of them (fewer centres than rows: every row its own)
    runs = 1 + int((np.diff(o) < 0).sum())
    assert runs <= min(m, n)
    if n >= 4096 and d >= 5:
        changes = int((comp[o][1:] != comp[o][:-1]).sum())
        assert changes <= 2 * m + 4, changes              # as drawn: ~ n (1 - 1 / kc)
    assert nat.cluster_rows(Yd, m).shape == Yd.shape


@pytest.mark.parametrize('d,k', [(8, 6), (32, 16)])
def test_gmm_fullcov_rows_stored_cluster_by_cluster(nat, ws, gpu_device, d, k):
    """Well-separated components, the rows once as drawn and once in the order the host layer stores them
    (`cluster_row_order`: neighbouring rows from the same component).  In that order nearly every (component, 32-row
    half) block of the raw-sums phase has no responsibility above 2^-40 and is skipped; both orders must give the
    float64 oracle's sums -- including a component that owns no row at all and one chain whose components sit far
    from the data (nothing is skipped for it)."""
    from bayes_vi_amd.fused_target import cluster_row_order
    rng = np.random.default_rng(31 * d + k)
    n, c = 6000, 3
    true_mu = rng.integers(-20, 21, (k, d)).astype(np.float64)
    comp = rng.integers(0, k - 1, n)                       # the last component owns no row
    Y = (true_mu[comp] + rng.normal(size=(n, d))).astype(np.float32)
    w = rng.dirichlet(np.ones(k) * 5, c).astype(np.float32)
    mu = (true_mu[None] + rng.normal(0, 0.2, (c, k, d))).astype(np.float32)
    mu[2] += 3.0                                           # every row a few widths from every component of chain 2
    L = np.tril(rng.normal(size=(c, k, d, d)) * 0.05)
    for i in range(d):
        L[..., i, i] = rng.uniform(0.9, 1.3, (c, k))
    L = L.astype(np.float32)
    theta = np.concatenate([w, mu.reshape(c, -1), L.reshape(c, -1)], -1)
    ref_lp, (gw, gmu, gL) = cf.gmm_fullcov(Y, w, mu, L)
    ref_g = np.concatenate([gw, gmu.reshape(c, -1), gL.reshape(c, -1)], -1)
    Yd = dev(Y, gpu_device)
    order = cluster_row_order(Yd, k)
    assert torch.equal(torch.sort(order).values, torch.arange(n, device=gpu_device))
    changes = int((torch.tensor(comp, device=gpu_device)[order][1:] != torch.tensor(comp, device=gpu_device)[order][:-1]).sum())
    assert changes < 8 * k                                 # the order does group the rows by component
    for rows in (Yd, Yd[order].contiguous()):
        lp, g = nat.gmm_fullcov_logp_grad(rows, dev(theta, gpu_device), k, ws)
        assert_logp_close(lp.cpu().numpy(), ref_lp)
        assert_grad_close(g.cpu().numpy(), ref_g, term_scale=np.full_like(ref_g, 3.0 * n))


@pytest.mark.parametrize('d,k,n,spread,pair_gap', [(32, 16, 6000 + 17, 1.0, None), (8, 2, 4096 + 5, 1.0, None),
                                                     (5, 3, 3000, 0.7, None), (16, 4, 5000, 1.0, None),
                                                     (32, 16, 8000, 1.0, 3.0), (8, 6, 5003, 1.0, 2.5),
                                                     (32, 12, 70, 1.0, None)])
def test_gmm_fullcov_tile_kinds(nat, ws, gpu_device, d, k, n, spread, pair_gap):
    """The three ways a 64-row tile is evaluated, against the float64 oracle, on rows stored cluster by cluster:
    tiles one component reaches (a kernel of their own: no log-sum-exp; with K <= 4 the components are replicated over
    the waves; a row count that is not a multiple of 64 puts rows of weight 0 into its last tile), tiles several
    components reach with some ruled out (`pair_gap`: the components come in pairs a few widths apart, so every tile
    is reached by two of them), and tiles every component reaches (a chain whose components all sit on the data's
    centre).  One chain has a non-finite parameter: its outputs are non-finite, the others' are untouched."""
    from bayes_vi_amd.fused_target import cluster_row_order
    rng = np.random.default_rng(1000 * d + 10 * k + n % 97)
    true_mu = rng.integers(-20, 21, (k, d)).astype(np.float64)
    if pair_gap is not None:
        for j in range(1, k, 2):
            true_mu[j] = true_mu[j - 1] + pair_gap * spread * rng.normal(size=d) / np.sqrt(d)
    comp = rng.integers(0, k, n)
    Y = (true_mu[comp] + spread * rng.normal(size=(n, d))).astype(np.float32)
    c = 4
    w = rng.dirichlet(np.ones(k) * 5, c).astype(np.float32)
    mu = (true_mu[None] + rng.normal(0, 0.1 * spread, (c, k, d))).astype(np.float32)
    mu[2] = Y.mean(0) + rng.normal(0, 0.05, (k, d))           # chain 2: every component in the middle of everything
    L = np.tril(rng.normal(size=(c, k, d, d)) * 0.03 * spread)
    for i in range(d):
        L[..., i, i] = spread * rng.uniform(0.9, 1.2, (c, k))
    L[2] *= 8.0
    L = L.astype(np.float32)
    theta = np.concatenate([w, mu.reshape(c, -1), L.reshape(c, -1)], -1)
    ref_lp, (gw, gmu, gL) = cf.gmm_fullcov(Y, w, mu, L)
    ref_g = np.concatenate([gw, gmu.reshape(c, -1), gL.reshape(c, -1)], -1)
    theta[3, k + 1] = np.nan                                   # chain 3: one mean is not a number
    Yd = dev(Y, gpu_device)
    rows = Yd[cluster_row_order(Yd, k)].contiguous() if n >= 256 else Yd
    lp, g = nat.gmm_fullcov_logp_grad(rows, dev(theta, gpu_device), k, ws)
    lp, g = lp.cpu().numpy(), g.cpu().numpy()
    assert_logp_close(lp[:3], ref_lp[:3])
    assert_grad_close(g[:3], ref_g[:3], term_scale=np.full_like(ref_g[:3], 3.0 * n))
    assert not np.isfinite(lp[3])
    lpv, none = nat.gmm_fullcov_logp_grad(rows, dev(theta, gpu_device), k, ws, want_grad=False)
    assert none is None
    np.testing.assert_array_equal(lpv.cpu().numpy()[:3], lp[:3])
    # the same call again: bit-identical (fixed order of every sum)
    lp2, g2 = nat.gmm_fullcov_logp_grad(rows, dev(theta, gpu_device), k, ws)
    assert np.array_equal(lp2.cpu().numpy()[:3], lp[:3]) and np.array_equal(g2.cpu().numpy()[:3], g[:3])


@pytest.mark.parametrize('d,k', [(32, 8), (6, 4)])
def test_gmm_fullcov_bounds_hold_for_hard_scale_matrices(nat, ws, gpu_device, d, k):
    """What the per-tile bounds must not get wrong, against the float64 oracle on rows stored cluster by cluster and
    packed: scale matrices with condition number ~1000 (a bound from the distance to the mean alone overshoots along
    the long axis: the component must stay wherever it matters), one broad component that covers every cluster (it is
    within reach of every tile and carries a few percent of every row), a component of weight 1e-30 (out of reach
    everywhere but finite), strongly correlated factors, and chains whose means sit between clusters."""
    rng = np.random.default_rng(77 * d + k)
    n, c = 7000, 5
    true_mu = rng.integers(-15, 16, (k, d)).astype(np.float64)
    comp = rng.integers(0, k, n)
    Y = (true_mu[comp] + rng.normal(size=(n, d))).astype(np.float32)
    w = rng.dirichlet(np.ones(k) * 5, c)
    mu = true_mu[None] + rng.normal(0, 0.1, (c, k, d))
    L = np.zeros((c, k, d, d))
    for ci in range(c):
        for kk in range(k):
            q, _ = np.linalg.qr(rng.normal(size=(d, d)))
            ev = np.ones(d)
            if ci == 0:
                ev = np.geomspace(0.05, 50.0, d)              # condition number 1000 in every component
            elif ci == 1 and kk == 0:
                ev = np.full(d, 20.0)                         # one broad component over everything
            elif ci == 2:
                ev = rng.uniform(0.5, 2.0, d)
                ev[0] = 30.0                                  # one long axis per component
            cov = (q * ev ** 2) @ q.T
            L[ci, kk] = np.linalg.cholesky(cov)
    w[3, 1] = 1e-30                                           # chain 3: a component of (almost) no weight
    mu[4] = 0.5 * (true_mu + np.roll(true_mu, 1, 0))[None]    # chain 4: every mean half way between two clusters
    w, mu, L = w.astype(np.float32), mu.astype(np.float32), L.astype(np.float32)
    theta = np.concatenate([w, mu.reshape(c, -1), L.reshape(c, -1)], -1)
    ref_lp, (gw, gmu, gL) = cf.gmm_fullcov(Y, w, mu, L)
    ref_g = np.concatenate([gw, gmu.reshape(c, -1), gL.reshape(c, -1)], -1)
    rows = nat.cluster_rows(dev(Y, gpu_device), 4 * k)
    packed = nat.gmm_fullcov_pack(rows)
    lp, g = nat.gmm_fullcov_logp_grad(rows, dev(theta, gpu_device), k, ws, packed=packed)
    assert_logp_close(lp.cpu().numpy(), ref_lp)
    assert_grad_close(g.cpu().numpy(), ref_g, term_scale=np.full_like(ref_g, 3.0 * n))
    lp0, g0 = nat.gmm_fullcov_logp_grad(dev(Y, gpu_device), dev(theta, gpu_device), k, ws)      # rows as drawn, unpacked
    assert_logp_close(lp0.cpu().numpy(), ref_lp)
    assert_grad_close(g0.cpu().numpy(), ref_g, term_scale=np.full_like(ref_g, 3.0 * n))


def test_gmm_fullcov_random_shapes(nat, ws, gpu_device):
    """Thirty random shapes (K 1 .. 16, D 1 .. 32, 1 .. 2500 rows, 1 .. 5 chains; separated, overlapping or coincident
    components; rows as drawn or cluster by cluster) against the float64 oracle: whatever mix of tile kinds the masks
    produce, the sums are the oracle's."""
    from bayes_vi_amd.fused_target import cluster_row_order
    rng = np.random.default_rng(20261112)
    for trial in range(30):
        k, d = int(rng.integers(1, 17)), int(rng.integers(1, 33))
        n, c = int(rng.integers(1, 2501)), int(rng.integers(1, 6))
        sep = float(rng.choice([0.0, 1.5, 4.0, 12.0]))
        true_mu = rng.normal(size=(k, d)) * sep
        comp = rng.integers(0, k, n)
        Y = (true_mu[comp] + rng.normal(size=(n, d))).astype(np.float32)
        w = (rng.dirichlet(np.ones(k) * 3, c) + 1e-3).astype(np.float32)
        mu = (true_mu[None] + rng.normal(0, 0.2, (c, k, d))).astype(np.float32)
        L = np.tril(rng.normal(size=(c, k, d, d)) * rng.choice([0.0, 0.05, 0.2]))
        for i in range(d):
            L[..., i, i] = rng.uniform(0.7, 1.5, (c, k))
        L = L.astype(np.float32)
        theta = np.concatenate([w, mu.reshape(c, -1), L.reshape(c, -1)], -1)
        ref_lp, (gw, gmu, gL) = cf.gmm_fullcov(Y, w, mu, L)
        ref_g = np.concatenate([gw, gmu.reshape(c, -1), gL.reshape(c, -1)], -1)
        Yd = dev(Y, gpu_device)
        if trial % 2 == 0 and n >= 64:
            Yd = Yd[cluster_row_order(Yd, k)].contiguous()
        lp, g = nat.gmm_fullcov_logp_grad(Yd, dev(theta, gpu_device), k, ws)
        what = 'trial %d: K=%d D=%d n=%d C=%d sep=%g' % (trial, k, d, n, c, sep)
        try:
            assert_logp_close(lp.cpu().numpy(), ref_lp)
            assert_grad_close(g.cpu().numpy(), ref_g, term_scale=np.full_like(ref_g, 3.0 * n))
        except AssertionError as e:
            raise AssertionError(what + ': ' + str(e))
        lpv, _ = nat.gmm_fullcov_logp_grad(Yd, dev(theta, gpu_device), k, ws, want_grad=False)
        assert torch.equal(lpv, lp), what


def test_logp_rebase_and_reference_level(nat, ws, gpu_device):
    """bvi_logp_rebase: the largest finite stored log-density becomes 0, the level takes it (split over the row
    shards), the absolute values come back, non-finite entries stay what they are; and a likelihood entry point
    given that level returns log-densities relative to it (Normal family, against the absolute call)."""
    lp = torch.tensor([-1.25e7, float('nan'), -1.2500003e7, float('-inf'), -1.3e7], device=gpu_device)
    ref = torch.full((1,), 100.0, dtype=torch.float64, device=gpu_device)
    before = lp.clone()
    absolute = torch.empty_like(lp)
    nat.logp_rebase(lp, 4, ref, absolute)
    assert float(ref) == 100.0 + (-1.25e7) / 4
    assert lp[0] == 0 and lp[2] == before[2] - before[0] and lp[4] == before[4] - before[0]
    assert bool(torch.isnan(lp[1])) and lp[3] == float('-inf')
    fin = torch.tensor([0, 2, 4], device=gpu_device)
    assert torch.equal(absolute[fin], (before[fin].double() + 400.0).float())
    # nothing finite: the level stays
    bad = torch.tensor([float('nan'), float('-inf')], device=gpu_device)
    nat.logp_rebase(bad, 1, ref)
    assert float(ref) == 100.0 + (-1.25e7) / 4
    # 1500 chains (more than one pass of the workgroup over the chains)
    big = -1e6 - torch.arange(1500, device=gpu_device, dtype=torch.float32)
    ref.zero_()
    nat.logp_rebase(big, 1, ref)
    assert float(ref) == -1e6 and big[0] == 0 and big[1499] == -1499
    g = torch.Generator(device=gpu_device); g.manual_seed(2)
    y = torch.empty(200_001, device=gpu_device).normal_(2.0, 1.5, generator=g)
    th = torch.tensor([[2.0, 1.5], [2.1, 1.4], [1.9, 1.6]], device=gpu_device)
    lp_abs, g_abs = nat.normal_iid_logp_grad(y, th, ws)
    level = torch.tensor([float(lp_abs[0])], dtype=torch.float64, device=gpu_device)
    lp_rel, g_rel = nat.normal_iid_logp_grad(y, th, ws, logp_ref=level)
    assert torch.equal(g_abs, g_rel)
    assert float((lp_rel.double() + level - lp_abs.double()).abs().max()) <= 0.5 * 2.0 ** -5   # half a float step at 3.7e5
    assert abs(float(lp_rel[0])) <= 0.02


def test_logp_rebase_shard_uneven_row_shards_keep_differences_exact(nat, ws, gpu_device):
    """bvi_logp_rebase_shard on two row shards of very different sizes (70 % / 30 % of 3M rows): every shard's level
    moves by its share of the rows, so each shard's own outputs stay small (with an equal split of the level the
    smaller shard's floats would sit at ~0.2 x the total, 1e6, where they are 0.06 apart), the total level is the
    same on both shards, and differences between chains agree with float64 to 1e-2."""
    g = torch.Generator(device=gpu_device); g.manual_seed(12)
    n = 3_000_000
    y = torch.empty(n, device=gpu_device).normal_(2.0, 1.5, generator=g)
    cut = int(0.7 * n) // 4 * 4
    th = torch.tensor([[2.0, 1.5], [2.0005, 1.5004], [1.9996, 1.4997], [2.001, 1.499]], device=gpu_device)
    th2 = th + torch.tensor([3e-4, -2e-4], device=gpu_device)
    shards = [(y[:cut], cut / n), (y[cut:].clone(), (n - cut) / n)]
    levels = [torch.zeros(2, dtype=torch.float64, device=gpu_device) for _ in shards]

    def total(theta, relative):
        parts = [nat.normal_iid_logp_grad(ys, theta, ws, logp_ref=lv if relative else None)[0]
                 for (ys, _), lv in zip(shards, levels)]
        return parts[0] + parts[1], parts

    lp, _ = total(th, False)                        # bootstrap: absolute pass, then the level moves to the best chain
    for (ys, share), lv in zip(shards, levels):
        nat.logp_rebase_shard(lp.clone(), share, lv)
    assert float(levels[0][1]) == float(levels[1][1]) == float(lp.max())
    assert float(levels[0][0]) == pytest.approx(float(lp.max()) * cut / n, rel=1e-12)
    lp, parts = total(th, True)
    assert all(float(p.abs().max()) < 5e3 for p in parts)         # each shard's own floats are small: O(sqrt(rows))
    # a transition's worth of movement, re-levelled, evaluated again
    lp2, _ = total(th2, True)
    absolute = torch.empty_like(lp2)
    for (ys, share), lv in zip(shards, levels):
        nat.logp_rebase_shard(lp2.clone(), share, lv, absolute)
    lp3, parts3 = total(th2, True)
    assert float(lp3.max()) == pytest.approx(0.0, abs=2e-3)
    assert all(float(p.abs().max()) < 5e3 for p in parts3)
    y64 = y.double()

    def ref64(theta):
        t = theta.double()
        r = (y64[None, :] - t[:, :1]) / t[:, 1:]
        return (-0.5 * r * r).sum(-1) - n * torch.log(t[:, 1]) - 0.5 * n * math.log(2 * math.pi)
    r3 = ref64(th2)
    d = lp3.double() - (r3 - r3.max())
    assert float((d - d.mean()).abs().max()) <= 1e-2, float((d - d.mean()).abs().max())
    assert float((absolute.double() - r3).abs().max()) <= 0.5            # absolute floats: spacing 0.25 at 5e6
    assert float(levels[0][1]) == float(levels[1][1])
