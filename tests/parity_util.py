"""Shared tolerance policy of the parity tests.

north_star tolerance: log-prob and gradient within 1e-5 relative in fp32.
  - logp: |got - ref| <= RTOL * |ref|  (+ 1e-6 absolute for values near zero).
  - grad: a gradient entry is a sum of N signed terms, so entries near a
    stationary point are differences of large numbers; per chain the error is
    judged against the largest entry of that chain's gradient,
    max|got - ref| <= RTOL * max|ref_c|  (+ 1e-6 absolute).
There is no floor proportional to the size of the summed terms: ABS_TERM is 0 and every default-path
kernel meets the bar as it stands, at every size tested (up to 10M / 100M rows).  A test of an explicitly
selected alternative precision mode may pass `abs_term` (times the sum of |terms| the caller supplies as
`term_scale`) and says so where it does.
"""
import numpy as np

RTOL = 1e-5
ABS_TERM = 0.0


def assert_logp_close(got, ref, rtol=RTOL, term_scale=None, abs_term=ABS_TERM):
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    tol = rtol * np.abs(ref) + 1e-6
    if term_scale is not None:
        tol = tol + abs_term * np.asarray(term_scale)
    bad = np.abs(got - ref) > tol
    assert not bad.any(), 'logp mismatch: got %s ref %s relerr %s' % (
        got[bad][:4], ref[bad][:4], (np.abs(got - ref) / np.abs(ref))[bad][:4])


def assert_grad_close(got, ref, rtol=RTOL, term_scale=None, abs_term=ABS_TERM):
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    scale = np.abs(ref).max(axis=-1, keepdims=True)
    tol = rtol * scale + 1e-6
    if term_scale is not None:
        tol = tol + abs_term * np.asarray(term_scale)
    err = np.abs(got - ref)
    bad = err > tol
    assert not bad.any(), 'grad mismatch: worst err/scale %.3e (entries %d of %d)' % (
        (err / np.maximum(scale, 1e-30)).max(), bad.sum(), bad.size)
