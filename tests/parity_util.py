"""Shared tolerance policy of the parity tests.

north_star tolerance: log-prob and gradient within 1e-5 relative in fp32.
  - logp: |got - ref| <= RTOL * |ref|  (+ a few ulp of the largest term).
  - grad: a gradient entry is a sum of N signed terms, so entries near a
    stationary point are differences of large numbers; per chain the error is
    judged against the largest entry of that chain's gradient,
    max|got - ref| <= RTOL * max|ref_c|, plus ABS_TERM * (sum of |terms|) when
    the caller can supply that scale (the floor any f32 summation has).
"""
import numpy as np

RTOL = 1e-5
ABS_TERM = 3e-7


def assert_logp_close(got, ref, rtol=RTOL, term_scale=None):
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    tol = rtol * np.abs(ref) + 1e-6
    if term_scale is not None:
        tol = tol + ABS_TERM * np.asarray(term_scale)
    bad = np.abs(got - ref) > tol
    assert not bad.any(), 'logp mismatch: got %s ref %s relerr %s' % (
        got[bad][:4], ref[bad][:4], (np.abs(got - ref) / np.abs(ref))[bad][:4])


def assert_grad_close(got, ref, rtol=RTOL, term_scale=None):
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    scale = np.abs(ref).max(axis=-1, keepdims=True)
    tol = rtol * scale + 1e-6
    if term_scale is not None:
        tol = tol + ABS_TERM * np.asarray(term_scale)
    err = np.abs(got - ref)
    bad = err > tol
    assert not bad.any(), 'grad mismatch: worst err/scale %.3e (entries %d of %d)' % (
        (err / np.maximum(scale, 1e-30)).max(), bad.sum(), bad.size)
