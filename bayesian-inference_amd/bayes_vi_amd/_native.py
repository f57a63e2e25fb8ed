"""ctypes binding of libbvi_hip.so (the C ABI declared in include/bvi.h).

There is no CPU fallback: if the library is missing or an entry point fails the
call raises.  Tensors are torch CUDA float32; kernels are enqueued on torch's
current stream and nothing here synchronises.
"""
from __future__ import annotations

import ctypes
import os

import torch

FAMILY_NORMAL_IID, FAMILY_LINREG, FAMILY_GMM1D, FAMILY_GMM_FULLCOV, FAMILY_MVN_TRIL = 0, 1, 2, 3, 4
MVN_SCALE_TRIL, MVN_CHOL_PRECISION_SOLVE = 0, 1
PREC_AUTO, PREC_F32, PREC_BF16X3, PREC_BF16X6 = 0, 1, 2, 3
PACK_BF16X3, PACK_F16 = 0, 1

BIJ_IDENTITY, BIJ_SOFTPLUS, BIJ_EXP, BIJ_SOFTMAX_CENTERED, BIJ_FILL_SCALE_TRIL = 0, 1, 2, 3, 4
PRIOR_NONE, PRIOR_NORMAL, PRIOR_HALFNORMAL, PRIOR_DIRICHLET, PRIOR_WISHART_TRIL = 0, 1, 2, 3, 4
MAX_PARAM_BLOCKS = 16
NOALLOC_MAX_CHAINS = 256
ERR_TOO_LARGE = -4
ABI_VERSION = 112     # BVI_VERSION of include/bvi.h this binding was written against


class ParamBlock(ctypes.Structure):
    """struct bvi_param_block of include/bvi.h."""
    _fields_ = [('bijector', ctypes.c_int32), ('size', ctypes.c_int32), ('prior', ctypes.c_int32),
                ('a', ctypes.c_float), ('b', ctypes.c_float), ('dim', ctypes.c_int32), ('shift', ctypes.c_float)]


NUTS_ACTIVE, NUTS_BUILDING, NUTS_TURNED, NUTS_DIVERGED, NUTS_MOVED, NUTS_FORWARD = 1, 2, 4, 8, 16, 32
_NUTS_FLOAT_ARRAYS = ('z', 'lp', 'grad', 'q', 'p', 'gq', 'lpq', 'ql', 'pl', 'gl', 'qr', 'pr', 'gr', 'sub_q', 'sub_g',
                      'sub_lp', 'ck_q', 'ck_p', 'h0', 'log_w', 'sub_log_w', 'sum_acc', 'log_u_sub', 'se', 'energy', 'sub_energy')
_NUTS_INT_ARRAYS = ('n_leap', 'leaf_ctr', 'flags')


class NutsState(ctypes.Structure):
    """struct bvi_nuts_state of include/bvi.h."""
    _fields_ = ([('n_chains', ctypes.c_int), ('n_unconstrained', ctypes.c_int), ('max_tree_depth', ctypes.c_int),
                 ('step_size', ctypes.c_float), ('max_energy_diff', ctypes.c_float)]
                + [(n, ctypes.c_void_p) for n in _NUTS_FLOAT_ARRAYS + _NUTS_INT_ARRAYS])


_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'lib', 'libbvi_hip.so')
_lib = None


class BviError(RuntimeError):
    """A libbvi_hip entry point returned a non-zero status."""


def lib_path() -> str:
    return _LIB_PATH


def lib():
    """Load libbvi_hip.so once; raise (never fall back) when it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise BviError(
                'libbvi_hip.so not found at %s -- build it with `make -C bayesian-inference_amd/csrc` '
                '(or __graft_entry__.build()); there is no CPU fallback' % _LIB_PATH)
        l = ctypes.CDLL(_LIB_PATH)
        vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
        l.bvi_version.restype = i32
        if l.bvi_version() != ABI_VERSION:
            raise BviError('libbvi_hip.so at %s has ABI version %d, this binding needs %d -- rebuild it with '
                           '`make -C bayesian-inference_amd/csrc`' % (_LIB_PATH, l.bvi_version(), ABI_VERSION))
        l.bvi_status_string.restype = ctypes.c_char_p
        l.bvi_status_string.argtypes = [i32]
        l.bvi_workspace_bytes.argtypes = [i32, i64, i32, i32, i32, ctypes.POINTER(sz)]
        l.bvi_normal_iid_logp_grad.argtypes = [vp, i64, vp, i32, vp, vp, vp, vp, sz, vp]
        l.bvi_linreg_logp_grad.argtypes = [vp, vp, i64, i32, vp, i32, i32, vp, vp, vp, vp, sz, vp]
        l.bvi_normal_iid_logp_grad_noalloc.argtypes = [vp, i64, vp, i32, vp, vp, vp]
        l.bvi_normal_iid_logp_grad_noalloc.restype = i32
        l.bvi_gmm1d_logp_grad.argtypes = [vp, i64, i32, vp, i32, vp, vp, vp, vp, sz, vp]
        l.bvi_gmm_fullcov_logp_grad.argtypes = [vp, i64, i32, i32, vp, i32, vp, vp, vp, vp, sz, vp]
        l.bvi_gmm_fullcov_packed_bytes.argtypes = [i64, i32, ctypes.POINTER(sz)]
        l.bvi_gmm_fullcov_pack_scratch_bytes.argtypes = [i32, ctypes.POINTER(sz)]
        l.bvi_gmm_fullcov_pack.argtypes = [vp, i64, i32, vp, sz, vp, sz, vp]
        l.bvi_gmm_fullcov_packed_logp_grad.argtypes = [vp, vp, i64, i32, i32, vp, i32, vp, vp, vp, vp, sz, vp]
        l.bvi_mvn_tril_packed_logp_grad.argtypes = [vp, vp, i64, i32, i32, vp, i32, vp, vp, vp, vp, sz, vp]
        l.bvi_cluster_rows_scratch_bytes.argtypes = [i64, i32, i32, ctypes.POINTER(sz)]
        l.bvi_cluster_rows.argtypes = [vp, i64, i32, i32, vp, vp, vp, sz, vp]
        l.bvi_mvn_tril_logp_grad.argtypes = [vp, i64, i32, i32, vp, i32, vp, vp, vp, vp, sz, vp]
        l.bvi_logp_rebase.argtypes = [i32, vp, i32, vp, vp, vp]
        l.bvi_logp_rebase_shard.argtypes = [i32, vp, ctypes.c_double, vp, vp, vp]
        l.bvi_linreg_packed_bytes.argtypes = [i64, i32, i32, ctypes.POINTER(sz)]
        l.bvi_linreg_pack_scratch_bytes.argtypes = [i32, i32, ctypes.POINTER(sz)]
        l.bvi_linreg_pack.argtypes = [vp, vp, i64, i32, i32, vp, sz, vp, sz, vp]
        l.bvi_linreg_packed_logp_grad.argtypes = [vp, i64, i32, i32, vp, i32, i32, vp, vp, vp, vp, sz, vp]
        bp, u64, f32 = ctypes.POINTER(ParamBlock), ctypes.c_uint64, ctypes.c_float
        l.bvi_constrain.argtypes = [bp, i32, i32, vp, vp, vp]
        l.bvi_posterior_assemble.argtypes = [bp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
        l.bvi_leapfrog_pre.argtypes = [bp, i32, i32, f32, vp, vp, vp, vp, vp]
        l.bvi_leapfrog_post.argtypes = [bp, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp, vp]
        l.bvi_leapfrog_mid.argtypes = [bp, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp, vp]
        l.bvi_leapfrog_pre_chainwise.argtypes = [bp, i32, i32, vp, vp, vp, vp, vp, vp]
        l.bvi_leapfrog_post_chainwise.argtypes = [bp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        nsp = ctypes.POINTER(NutsState)
        l.bvi_nuts_begin.argtypes = [nsp, u64, u64, vp]
        l.bvi_nuts_subtree_begin.argtypes = [nsp, u64, u64, i32, vp]
        l.bvi_nuts_leaf.argtypes = [nsp, u64, u64, i32, vp, vp]
        l.bvi_nuts_subtree_end.argtypes = [nsp, vp, vp]
        l.bvi_hmc_begin.argtypes = [i32, i32, u64, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        l.bvi_hmc_end.argtypes = [i32, i32, u64, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        l.bvi_hmc_fused.argtypes = [i32, vp, vp, i64, i32, i32, bp, i32, i32, f32, vp, i32, i32, u64, u64, i32,
                                    vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        l.bvi_nuts_fused.argtypes = [i32, vp, vp, i64, i32, i32, bp, i32, i32, f32, vp, i32, f32, i32, u64, u64, i32,
                                     vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        l.bvi_rwm_fused.argtypes = [i32, vp, vp, i64, i32, i32, bp, i32, i32, f32, i32, u64, u64, i32,
                                    vp, vp, vp, vp, vp, vp, vp, vp]
        l.bvi_advi_meanfield_sample.argtypes = [i32, i32, vp, vp, vp, vp, vp]
        l.bvi_advi_meanfield_grads.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
        l.bvi_advi_fullrank_sample.argtypes = [i32, i32, vp, vp, vp, vp, vp]
        l.bvi_advi_fullrank_grads.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
        l.bvi_clip_adam_step.argtypes = [i32, vp, vp, vp, vp, vp, i32, f32, f32, f32, f32, f32, vp]
        l.bvi_linreg_recommended_format.argtypes = [i64, i32, i32, ctypes.POINTER(ctypes.c_int)]
        l.bvi_philox_uint32.argtypes = [u64, u64, i32, i32, i32, vp, vp]
        l.bvi_philox_normal.argtypes = [u64, u64, i32, i32, vp, vp]
        for name in ('bvi_constrain', 'bvi_posterior_assemble', 'bvi_leapfrog_pre', 'bvi_leapfrog_post',
                     'bvi_hmc_begin', 'bvi_hmc_end', 'bvi_hmc_fused', 'bvi_nuts_fused', 'bvi_rwm_fused', 'bvi_philox_uint32', 'bvi_philox_normal',
                     'bvi_leapfrog_mid', 'bvi_leapfrog_pre_chainwise', 'bvi_leapfrog_post_chainwise', 'bvi_nuts_begin',
                     'bvi_nuts_subtree_begin', 'bvi_nuts_leaf', 'bvi_nuts_subtree_end'):
            getattr(l, name).restype = i32
        for name in ('bvi_workspace_bytes', 'bvi_normal_iid_logp_grad', 'bvi_linreg_logp_grad',
                     'bvi_gmm1d_logp_grad', 'bvi_gmm_fullcov_logp_grad', 'bvi_mvn_tril_logp_grad', 'bvi_linreg_packed_bytes', 'bvi_linreg_pack', 'bvi_linreg_pack_scratch_bytes',
                     'bvi_linreg_packed_logp_grad', 'bvi_logp_rebase', 'bvi_logp_rebase_shard', 'bvi_linreg_recommended_format', 'bvi_advi_meanfield_sample', 'bvi_advi_meanfield_grads', 'bvi_advi_fullrank_sample',
                     'bvi_advi_fullrank_grads',
                     'bvi_clip_adam_step'):
            getattr(l, name).restype = i32
        _lib = l
    return _lib


def check(status: int, what: str):
    if status != 0:
        raise BviError('%s failed with status %d: %s' % (what, status, lib().bvi_status_string(status).decode()))


def workspace_bytes(family: int, n_rows: int, n_chains: int, d: int = 0, k: int = 0) -> int:
    out = ctypes.c_size_t(0)
    check(lib().bvi_workspace_bytes(family, n_rows, n_chains, d, k, ctypes.byref(out)), 'bvi_workspace_bytes')
    return int(out.value)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check_f32_cuda(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise BviError('libbvi_hip needs contiguous float32 CUDA tensors, got %s %s contiguous=%s'
                           % (t.device, t.dtype, t.is_contiguous()))


class Workspace:
    """A grow-only device scratch buffer owned by the caller of the C ABI."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, nbytes: int):
        nbytes = max(int(nbytes), 256)
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self.buf


def _ref_ptr(logp_ref):
    """Device address of the level the likelihood entry points subtract in double (one float64), or NULL."""
    if logp_ref is None:
        return None
    if not (logp_ref.is_cuda and logp_ref.dtype == torch.float64 and logp_ref.numel() in (1, 2)):
        raise BviError('logp_ref must be one float64 on the device (or the [own, total] pair of bvi_logp_rebase_shard)')
    return logp_ref.data_ptr()


def logp_rebase(lp, n_shards, logp_ref, lp_absolute=None):
    """bvi_logp_rebase: lp -= max finite lp; logp_ref += that / n_shards; lp_absolute <- the absolute values."""
    _check_f32_cuda(lp)
    check(lib().bvi_logp_rebase(lp.shape[0], _ptr(lp), int(n_shards), _ref_ptr(logp_ref), _ptr(lp_absolute), _stream()),
          'bvi_logp_rebase')


def logp_rebase_shard(lp, share, levels, lp_absolute=None):
    """bvi_logp_rebase_shard: lp -= m (its largest finite entry); levels[0] += m * share (this row shard's own level);
    levels[1] += m (the level of the total over all shards); lp_absolute <- the absolute values."""
    _check_f32_cuda(lp)
    if not (levels.is_cuda and levels.dtype == torch.float64 and levels.numel() == 2 and levels.is_contiguous()):
        raise BviError('levels must be two float64 on the device: [own level, total level]')
    check(lib().bvi_logp_rebase_shard(lp.shape[0], _ptr(lp), float(share), levels.data_ptr(), _ptr(lp_absolute),
                                      _stream()), 'bvi_logp_rebase_shard')


def _outputs(theta, want_grad):
    c = theta.shape[0]
    logp = torch.empty(c, dtype=torch.float32, device=theta.device)
    grad = torch.empty_like(theta) if want_grad else None
    return logp, grad


def normal_iid_logp_grad(y, theta, ws: Workspace, want_grad=True, logp_ref=None):
    _check_f32_cuda(y, theta)
    n, c = y.shape[0], theta.shape[0]
    assert theta.shape == (c, 2)
    logp, grad = _outputs(theta, want_grad)
    buf = ws.get(workspace_bytes(FAMILY_NORMAL_IID, n, c))
    check(lib().bvi_normal_iid_logp_grad(_ptr(y), n, _ptr(theta), c, _ptr(logp), _ptr(grad), _ref_ptr(logp_ref),
                                         _ptr(buf), buf.numel(), _stream()), 'bvi_normal_iid_logp_grad')
    return logp, grad


def linreg_logp_grad(X, y, theta, ws: Workspace, want_grad=True, precision=PREC_AUTO, logp_ref=None):
    _check_f32_cuda(X, y, theta)
    n, d = X.shape
    c = theta.shape[0]
    assert y.shape == (n,) and theta.shape == (c, d + 2)
    logp, grad = _outputs(theta, want_grad)
    buf = ws.get(workspace_bytes(FAMILY_LINREG, n, c, d=d))
    check(lib().bvi_linreg_logp_grad(_ptr(X), _ptr(y), n, d, _ptr(theta), c, precision, _ptr(logp), _ptr(grad),
                                     _ref_ptr(logp_ref), _ptr(buf), buf.numel(), _stream()), 'bvi_linreg_logp_grad')
    return logp, grad


def gmm1d_logp_grad(y, theta, k, ws: Workspace, want_grad=True, logp_ref=None):
    _check_f32_cuda(y, theta)
    n, c = y.shape[0], theta.shape[0]
    assert theta.shape == (c, 3 * k)
    logp, grad = _outputs(theta, want_grad)
    buf = ws.get(workspace_bytes(FAMILY_GMM1D, n, c, k=k))
    check(lib().bvi_gmm1d_logp_grad(_ptr(y), n, k, _ptr(theta), c, _ptr(logp), _ptr(grad), _ref_ptr(logp_ref),
                                    _ptr(buf), buf.numel(), _stream()), 'bvi_gmm1d_logp_grad')
    return logp, grad


def linreg_recommended_format(n, d, n_chains):
    """bvi_linreg_recommended_format: PACK_F16 when the packed f16-plane kernel is the route for this shape, else None."""
    out = ctypes.c_int(-1)
    check(lib().bvi_linreg_recommended_format(int(n), int(d), int(n_chains), ctypes.byref(out)),
          'bvi_linreg_recommended_format')
    return None if out.value < 0 else int(out.value)


def linreg_pack(X, y, fmt=PACK_BF16X3):
    """One-time split of (X, y) into the piece planes of a matrix-core kernel."""
    _check_f32_cuda(X, y)
    n, d = X.shape
    out = ctypes.c_size_t(0)
    check(lib().bvi_linreg_packed_bytes(n, d, fmt, ctypes.byref(out)), 'bvi_linreg_packed_bytes')
    packed = torch.empty(max(int(out.value), 16), dtype=torch.uint8, device=X.device)
    check(lib().bvi_linreg_pack_scratch_bytes(d, fmt, ctypes.byref(out)), 'bvi_linreg_pack_scratch_bytes')
    scratch = torch.empty(max(int(out.value), 16), dtype=torch.uint8, device=X.device)
    check(lib().bvi_linreg_pack(_ptr(X), _ptr(y), n, d, fmt, _ptr(packed), packed.numel(), _ptr(scratch),
                                scratch.numel(), _stream()), 'bvi_linreg_pack')
    return packed


def linreg_packed_logp_grad(packed, n, d, theta, ws: Workspace, want_grad=True, precision=PREC_AUTO,
                            fmt=PACK_BF16X3, logp_ref=None):
    _check_f32_cuda(theta)
    c = theta.shape[0]
    assert theta.shape == (c, d + 2)
    logp, grad = _outputs(theta, want_grad)
    buf = ws.get(workspace_bytes(FAMILY_LINREG, n, c, d=d))
    check(lib().bvi_linreg_packed_logp_grad(_ptr(packed), n, d, fmt, _ptr(theta), c, precision, _ptr(logp),
                                            _ptr(grad), _ref_ptr(logp_ref), _ptr(buf), buf.numel(), _stream()),
          'bvi_linreg_packed_logp_grad')
    return logp, grad


def make_blocks(specs):
    """specs: list of (bijector, size, prior, a, b[, dim, shift]) -> ctypes array of bvi_param_block."""
    if len(specs) > MAX_PARAM_BLOCKS:
        raise BviError('at most %d parameter blocks on the device path' % MAX_PARAM_BLOCKS)
    arr = (ParamBlock * len(specs))()
    for i, spec in enumerate(specs):
        bij, size, prior, a, b = spec[:5]
        dim, shift = spec[5:7] if len(spec) > 5 else (0, 0.0)
        arr[i] = ParamBlock(bij, size, prior, a, b, dim, shift)
    return arr


def philox_uint32(seed, step, stream_id, n_chains, n_blocks, device):
    out = torch.empty(n_chains, n_blocks, 4, dtype=torch.int32, device=device)
    check(lib().bvi_philox_uint32(seed, step, stream_id, n_chains, n_blocks, _ptr(out), _stream()), 'bvi_philox_uint32')
    return out


def philox_normal(seed, step, n_chains, n_params, device):
    out = torch.empty(n_chains, n_params, dtype=torch.float32, device=device)
    check(lib().bvi_philox_normal(seed, step, n_chains, n_params, _ptr(out), _stream()), 'bvi_philox_normal')
    return out


def gmm_fullcov_logp_grad(Y, theta, k, ws: Workspace, want_grad=True, logp_ref=None, packed=None):
    """bvi_gmm_fullcov_logp_grad, or bvi_gmm_fullcov_packed_logp_grad with `packed` = gmm_fullcov_pack(Y)."""
    _check_f32_cuda(Y, theta)
    n, d = Y.shape
    c = theta.shape[0]
    assert theta.shape == (c, k + k * d + k * d * d)
    logp, grad = _outputs(theta, want_grad)
    buf = ws.get(workspace_bytes(FAMILY_GMM_FULLCOV, n, c, d=d, k=k))
    if packed is not None:
        check(lib().bvi_gmm_fullcov_packed_logp_grad(_ptr(Y), _ptr(packed), n, d, k, _ptr(theta), c, _ptr(logp),
                                                     _ptr(grad), _ref_ptr(logp_ref), _ptr(buf), buf.numel(), _stream()),
              'bvi_gmm_fullcov_packed_logp_grad')
        return logp, grad
    check(lib().bvi_gmm_fullcov_logp_grad(_ptr(Y), n, d, k, _ptr(theta), c, _ptr(logp), _ptr(grad),
                                          _ref_ptr(logp_ref), _ptr(buf), buf.numel(), _stream()),
          'bvi_gmm_fullcov_logp_grad')
    return logp, grad


CLUSTER_MAX_CENTRES = 128


def gmm_fullcov_pack(Y):
    """bvi_gmm_fullcov_pack: what the full-covariance mixture / multivariate-normal entry points need of the rows
    alone (statistics, tile spheres, f16 pieces), once per dataset."""
    _check_f32_cuda(Y)
    n, d = Y.shape
    out = ctypes.c_size_t(0)
    check(lib().bvi_gmm_fullcov_packed_bytes(n, d, ctypes.byref(out)), 'bvi_gmm_fullcov_packed_bytes')
    packed = torch.empty(max(int(out.value), 16), dtype=torch.uint8, device=Y.device)
    check(lib().bvi_gmm_fullcov_pack_scratch_bytes(d, ctypes.byref(out)), 'bvi_gmm_fullcov_pack_scratch_bytes')
    scratch = torch.empty(max(int(out.value), 16), dtype=torch.uint8, device=Y.device)
    check(lib().bvi_gmm_fullcov_pack(_ptr(Y), n, d, _ptr(packed), packed.numel(), _ptr(scratch), scratch.numel(),
                                     _stream()), 'bvi_gmm_fullcov_pack')
    return packed


def cluster_rows(Y, n_centres, want_order=False):
    """bvi_cluster_rows: a copy of Y[n, d] with rows that lie close together stored next to each other (and, if asked
    for, order[n] with copy[i] = Y[order[i]])."""
    _check_f32_cuda(Y)
    n, d = Y.shape
    m = int(max(1, min(CLUSTER_MAX_CENTRES, n_centres)))
    out = ctypes.c_size_t(0)
    check(lib().bvi_cluster_rows_scratch_bytes(n, d, m, ctypes.byref(out)), 'bvi_cluster_rows_scratch_bytes')
    scratch = torch.empty(max(int(out.value), 16), dtype=torch.uint8, device=Y.device)
    Yout = torch.empty_like(Y)
    order = torch.empty(n, dtype=torch.int64, device=Y.device) if want_order else None
    check(lib().bvi_cluster_rows(_ptr(Y), n, d, m, _ptr(Yout), _ptr(order), _ptr(scratch), scratch.numel(),
                                 _stream()), 'bvi_cluster_rows')
    return (Yout, order) if want_order else Yout


def mvn_tril_logp_grad(Y, theta, parameterisation, ws: Workspace, want_grad=True, logp_ref=None, packed=None):
    """bvi_mvn_tril_logp_grad: Y[n, d], theta[C, d + d*d] = (loc, M) with M the scale (MVN_SCALE_TRIL) or a Cholesky
    factor of the precision as the reference passes it (MVN_CHOL_PRECISION_SOLVE)."""
    _check_f32_cuda(Y, theta)
    n, d = Y.shape
    c = theta.shape[0]
    assert theta.shape == (c, d + d * d)
    logp, grad = _outputs(theta, want_grad)
    buf = ws.get(workspace_bytes(FAMILY_MVN_TRIL, n, c, d=d))
    if packed is not None:
        check(lib().bvi_mvn_tril_packed_logp_grad(_ptr(Y), _ptr(packed), n, d, int(parameterisation), _ptr(theta), c,
                                                  _ptr(logp), _ptr(grad), _ref_ptr(logp_ref), _ptr(buf), buf.numel(),
                                                  _stream()), 'bvi_mvn_tril_packed_logp_grad')
        return logp, grad
    check(lib().bvi_mvn_tril_logp_grad(_ptr(Y), n, d, int(parameterisation), _ptr(theta), c, _ptr(logp), _ptr(grad),
                                       _ref_ptr(logp_ref), _ptr(buf), buf.numel(), _stream()), 'bvi_mvn_tril_logp_grad')
    return logp, grad
