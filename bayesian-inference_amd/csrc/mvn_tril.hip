// Single multivariate-normal likelihood y_i ~ N(loc, S S^T) for all chains: the one-component case of the
// full-covariance mixture kernels (gmm_fullcov.hip) plus the per-chain parameter maps around it.
//
// Reference: exploratory_experiments/multivariate-gaussian-test.ipynb cell 15,
//     tfd.MultivariateNormalTriL(loc=loc, scale_tril=tf.linalg.cholesky_solve(precision, tf.linalg.eye(dims)))
// with `precision` the Cholesky factor P of a Wishart draw.  cholesky_solve(P, I) is the FULL matrix
// Sigma = (P P^T)^-1, and MultivariateNormalTriL reads only the lower triangle of what it is given, so the scale the
// reference evaluates is S = tril(Sigma).  BVI_MVN_CHOL_PRECISION_SOLVE reproduces exactly that; BVI_MVN_SCALE_TRIL
// takes S itself as the parameter.
//
// Per chain, one 64-thread workgroup, double precision in LDS (d <= 32):
//   prepare:  Q = P^-1 (forward substitution, one column per thread), Sigma = Q^T Q, S = tril(Sigma);
//             kernel parameters (weight 1, loc, S) for the mixture kernels
//   gradient: with G = d logp / d S (lower triangular, from the mixture kernels' finalize) every entry of Sigma on or
//             below the diagonal feeds S once, so d logp / d Sigma = G (zero above the diagonal) and, from
//             d Sigma = -Sigma (dP P^T + P dP^T) Sigma,
//                 d logp / d P = -tril( Sigma (G + G^T) Sigma P ).
#include "bvi_common.h"

namespace {

constexpr int kMaxD = 32;
constexpr int kLd = kMaxD + 1;      // padded row length of the LDS matrices

__global__ __launch_bounds__(64) void mvn_prepare_kernel(int d, int mode, const float* __restrict__ theta, int C,
                                                         float* __restrict__ kth, double* __restrict__ sig) {
  const int c = blockIdx.x, t = threadIdx.x, dd = d * d;
  __shared__ double P[kMaxD * kLd], Q[kMaxD * kLd];
  const float* th = theta + (int64_t)c * (d + dd);
  float* out = kth + (int64_t)c * (1 + d + dd);
  if (t == 0) out[0] = 1.f;
  for (int j = t; j < d; j += 64) out[1 + j] = th[j];
  if (mode == BVI_MVN_SCALE_TRIL) {
    for (int e = t; e < dd; e += 64) out[1 + d + e] = (e % d <= e / d) ? th[d + e] : 0.f;
    return;
  }
  for (int e = t; e < dd; e += 64) P[(e / d) * kLd + e % d] = (e % d <= e / d) ? (double)th[d + e] : 0.0;
  __syncthreads();
  if (t < d) {                       // column t of Q = P^-1
    for (int i = 0; i < d; ++i) {
      double s = i == t ? 1.0 : 0.0;
      for (int k = t; k < i; ++k) s -= P[i * kLd + k] * Q[k * kLd + t];
      Q[i * kLd + t] = i < t ? 0.0 : s / P[i * kLd + i];
    }
  }
  __syncthreads();
  double* sg = sig + (int64_t)c * dd;
  for (int e = t; e < dd; e += 64) {
    const int i = e / d, j = e % d;
    double s = 0.0;
    for (int k = i > j ? i : j; k < d; ++k) s += Q[k * kLd + i] * Q[k * kLd + j];
    sg[e] = s;
    out[1 + d + e] = j <= i ? (float)s : 0.f;
  }
}

__global__ __launch_bounds__(64) void mvn_grad_kernel(int d, int mode, const float* __restrict__ theta,
                                                      const double* __restrict__ sig, const float* __restrict__ gk,
                                                      int C, float* __restrict__ grad) {
  const int c = blockIdx.x, t = threadIdx.x, dd = d * d;
  const float* g = gk + (int64_t)c * (1 + d + dd);
  float* out = grad + (int64_t)c * (d + dd);
  for (int j = t; j < d; j += 64) out[j] = g[1 + j];
  if (mode == BVI_MVN_SCALE_TRIL) {
    for (int e = t; e < dd; e += 64) out[d + e] = (e % d <= e / d) ? g[1 + d + e] : 0.f;
    return;
  }
  __shared__ double S[kMaxD * kLd], A[kMaxD * kLd], B[kMaxD * kLd];
  const double* sg = sig + (int64_t)c * dd;
  const float* gl = g + 1 + d;
  for (int e = t; e < dd; e += 64) {
    const int i = e / d, j = e % d;
    S[i * kLd + j] = sg[e];
    const double lo = j <= i ? (double)gl[i * d + j] : 0.0, up = i <= j ? (double)gl[j * d + i] : 0.0;
    A[i * kLd + j] = lo + up;                                // G + G^T
  }
  __syncthreads();
  for (int e = t; e < dd; e += 64) {                         // B = Sigma (G + G^T)
    const int i = e / d, j = e % d;
    double s = 0.0;
    for (int k = 0; k < d; ++k) s += S[i * kLd + k] * A[k * kLd + j];
    B[i * kLd + j] = s;
  }
  __syncthreads();
  for (int e = t; e < dd; e += 64) {                         // A = B Sigma
    const int i = e / d, j = e % d;
    double s = 0.0;
    for (int k = 0; k < d; ++k) s += B[i * kLd + k] * S[k * kLd + j];
    A[i * kLd + j] = s;
  }
  __syncthreads();
  const float* P = theta + (int64_t)c * (d + dd) + d;
  for (int e = t; e < dd; e += 64) {
    const int i = e / d, j = e % d;
    double s = 0.0;
    if (j <= i)
      for (int k = j; k < d; ++k) s += A[i * kLd + k] * (double)P[k * d + j];
    out[d + e] = (float)-s;
  }
}

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

size_t bvi_ws_mvn_tril(int64_t n, int c, int d) {
  const size_t row = (size_t)(1 + d + d * d) * sizeof(float);
  return align256(2 * (size_t)c * row) + align256((size_t)c * d * d * sizeof(double)) + bvi_ws_gmm_fullcov(n, c, d, 1);
}

int bvi_launch_mvn_tril(const float* Y, int64_t n, int d, int mode, const float* theta, int c, float* logp, float* grad,
                        float* ws, hipStream_t s, const double* logp_ref, const void* packed) {
  const size_t row = (size_t)(1 + d + d * d);
  unsigned char* base = reinterpret_cast<unsigned char*>(ws);
  float* kth = reinterpret_cast<float*>(base);
  float* kg = kth + (size_t)c * row;
  double* sig = reinterpret_cast<double*>(base + align256(2 * (size_t)c * row * sizeof(float)));
  float* inner = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(sig) + align256((size_t)c * d * d * sizeof(double)));
  hipLaunchKernelGGL(mvn_prepare_kernel, dim3(c), dim3(64), 0, s, d, mode, theta, c, kth, sig);
  const int st = bvi_launch_gmm_fullcov(Y, n, d, 1, kth, c, logp, grad ? kg : nullptr, inner, s, logp_ref, packed);
  if (st != BVI_OK) return st;
  if (grad) hipLaunchKernelGGL(mvn_grad_kernel, dim3(c), dim3(64), 0, s, d, mode, theta, sig, kg, c, grad);
  return (int)hipGetLastError();
}
