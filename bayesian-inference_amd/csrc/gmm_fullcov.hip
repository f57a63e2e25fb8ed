// Full-covariance Gaussian mixture likelihood and gradient for every chain:
//   y_i ~ sum_k w~_k N(mu_k, L_k L_k^T),  w~ = w / sum w,  L_k lower triangular.
// Replaces the likelihood node MixtureSameFamily(Categorical(probs),
// MultivariateNormalTriL(locs[K,D], scale_tril[K,D,D])) of
// exploratory_experiments/multivariate-gaussian-mixture-density-estimation-test.ipynb
// cell 7 (under tfd.Sample, bayes_vi/model/model.py:17-21) and its gradient.
//
// Maths per chain and row, with A_k = L_k^-1, z_ik = A_k y_i - A_k mu_k:
//   a_ik = log w~_k - sum_d log L_k,dd - D/2 log 2pi - |z_ik|^2 / 2
//   logp += LSE_k a_ik,   rho_ik = softmax_k a_ik
//   raw sums  S0_k = sum_i rho_ik,  S1_k = sum_i rho_ik z_ik,  S2_k = sum_i rho_ik z_ik z_ik^T
//   dlogp/dmu_k = A_k^T S1_k
//   dlogp/dL_k  = tril(A_k^T S2_k) - S0_k diag(1 / L_k,dd)
//   dlogp/dw_k  = S0_k / w_k - n / sum_j w_j
//
// Three launches: `prepare` (A_k, A_k mu_k and the constant of a_ik per chain and component,
// by forward substitution in double), the streaming kernel (exact f32 on
// v_mfma_f32_16x16x4_f32), `finalize` (fixed-order reduction of the row-block partials in
// double and the three matrix products above).
//
// Streaming kernel: one workgroup = one chain x a strided set of 16-row tiles; its 4 waves
// split the components (wave w owns k = w, w+4, ...).  A_k of the chain sits in LDS.
// Per tile: pass 1 computes z^T tiles (dims in registers, rows on lanes) so that |z|^2 is a
// register sum, and publishes a_ik through LDS for the log-sum-exp over k; pass 2 recomputes
// z in the other orientation (rows in registers, dims on lanes) -- the same MFMAs with the
// operands swapped -- where the accumulator registers are, lane for lane, BOTH operands of
// S2 += (rho z)^T z, and S0/S1 are register sums.
#include "bvi_common.h"

namespace {

constexpr int DP = 32;        // padded dimension
constexpr int LS = 34;        // LDS row stride in floats: conflict-free ds_read_b32 operand reads
constexpr int TR = 16;        // rows per tile

__global__ void fullcov_prepare(const float* __restrict__ theta, int C, int K, int D, float* __restrict__ Ainv,
                                float* __restrict__ bvec, float* __restrict__ cst) {
  // one block of DP threads per (chain, component)
  const int c = blockIdx.x / K, k = blockIdx.x % K, j = threadIdx.x;
  const int P = K + K * D + K * D * D;
  const float* th = theta + (int64_t)c * P;
  const float* mu = th + K + k * D;
  const float* L = th + K + K * D + k * D * D;
  __shared__ double A[DP][DP + 1];
  for (int i = 0; i < DP; ++i) A[i][j] = 0.0;
  if (j < D) {
    // column j of L^-1 by forward substitution
    for (int i = j; i < D; ++i) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int m = j; m < i; ++m) s -= (double)L[i * D + m] * A[m][j];
      A[i][j] = s / (double)L[i * D + i];
    }
  }
  __syncthreads();
  float* Ao = Ainv + ((int64_t)c * K + k) * DP * DP;
  for (int i = 0; i < DP; ++i) Ao[i * DP + j] = (float)A[i][j];
  double b = 0.0;
  if (j < D)
    for (int m = 0; m < D; ++m) b += A[j][m] * (double)mu[m];
  bvec[((int64_t)c * K + k) * DP + j] = (float)b;
  if (j == 0) {
    double wsum = 0.0, ld = 0.0;
    for (int m = 0; m < K; ++m) wsum += (double)th[m];
    for (int m = 0; m < D; ++m) ld += log((double)L[m * D + m]);
    cst[c * K + k] = (float)(log((double)th[k]) - log(wsum) - ld - 0.5 * D * BVI_LN2PI_D);
  }
}

template <int KQ>  // components per wave (K <= 4 * KQ)
__global__ __launch_bounds__(256) void fullcov_kernel(const float* __restrict__ Y, int64_t n, int D, int K,
                                                      const float* __restrict__ Ainv,
                                                      const float* __restrict__ bvec,
                                                      const float* __restrict__ cst, int C,
                                                      float* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                         // [K][DP][LS]
  float* bs = As + K * DP * LS;            // [K][DP]
  float* ys = bs + K * DP;                 // [TR][LS]
  float* al = ys + TR * LS;                // [4*KQ][TR]
  float* ls = al + 4 * KQ * TR;            // [TR] log-sum-exp per row (+inf for rows past the end)
  float* cs = ls + TR;                     // [4*KQ]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int c = blockIdx.y;
  for (int idx = tid; idx < K * DP * DP; idx += 256) {
    const int k = idx / (DP * DP), r = (idx / DP) % DP, col = idx % DP;
    As[(k * DP + r) * LS + col] = Ainv[((int64_t)c * K + k) * DP * DP + r * DP + col];
  }
  for (int idx = tid; idx < K * DP; idx += 256) bs[idx] = bvec[(int64_t)c * K * DP + idx];
  for (int idx = tid; idx < 4 * KQ; idx += 256) cs[idx] = idx < K ? cst[c * K + idx] : -INFINITY;

  f32x4 S2[KQ][2][2];
  float S1[KQ][2], S0[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    S0[q] = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      S1[q][a] = 0.f;
#pragma unroll
      for (int b = 0; b < 2; ++b) S2[q][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  float logp_acc = 0.f;

  const int64_t ntiles = (n + TR - 1) / TR;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    __syncthreads();   // previous tile fully consumed (and the LDS prologue complete)
    {
      // stage the tile: thread -> (row, 2 features); zeros past n or D
      const int row = tid >> 4, col = (tid & 15) * 2;
      const int64_t gr = t * TR + row;
#pragma unroll
      for (int e = 0; e < 2; ++e)
        ys[row * LS + col + e] = (gr < n && col + e < D) ? Y[gr * D + col + e] : 0.f;
    }
    __syncthreads();
    // ---- pass 1: a_ik for this wave's components ----
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int k = wave + 4 * q;
      if (k < K) {
        float nrm = 0.f;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          f32x4 acc;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = -bs[k * DP + 16 * dt + 4 * g + r];
#pragma unroll
          for (int ks = 0; ks < 8; ++ks)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(As[(k * DP + 16 * dt + li) * LS + 4 * ks + g],
                                                       ys[li * LS + 4 * ks + g], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) nrm = fmaf(acc[r], acc[r], nrm);
        }
        nrm += __shfl_xor(nrm, 16, 64);
        nrm += __shfl_xor(nrm, 32, 64);
        if (g == 0) al[k * TR + li] = cs[k] - 0.5f * nrm;
      }
    }
    __syncthreads();
    if (tid < TR) {
      float m = -INFINITY;
      for (int k = 0; k < K; ++k) m = fmaxf(m, al[k * TR + tid]);
      float s = 0.f;
      for (int k = 0; k < K; ++k) s += __expf(al[k * TR + tid] - m);
      const float lse = m + __logf(s);
      const bool valid = t * TR + tid < n;
      if (valid) logp_acc += lse;
      ls[tid] = valid ? lse : INFINITY;
    }
    __syncthreads();
    // ---- pass 2: z with rows in registers, responsibilities, raw sums ----
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int k = wave + 4 * q;
      if (k < K) {
        f32x4 z[2], rho, rz[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) rho[r] = __expf(al[k * TR + 4 * g + r] - ls[4 * g + r]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const float bb = -bs[k * DP + 16 * dt + li];
          z[dt] = (f32x4){bb, bb, bb, bb};
#pragma unroll
          for (int ks = 0; ks < 8; ++ks)
            z[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ys[li * LS + 4 * ks + g],
                                                         As[(k * DP + 16 * dt + li) * LS + 4 * ks + g], z[dt], 0,
                                                         0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            rz[dt][r] = rho[r] * z[dt][r];
            S1[q][dt] += rz[dt][r];
          }
        }
        S0[q] += rho[0] + rho[1] + rho[2] + rho[3];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              S2[q][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(rz[a][r], z[b][r], S2[q][a][b], 0, 0, 0);
      }
    }
  }

  // ---- partials of this (row block, chain): [K][1 + DP + DP*DP] then the log-prob ----
  constexpr int Q = 1 + DP + DP * DP;
  float* out = partials + ((int64_t)blockIdx.x * C + c) * ((int64_t)K * Q + 1);
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int k = wave + 4 * q;
    if (k < K) {
      float s0 = S0[q];   // every lane of a 16-lane group holds the same rows: take one, add the 4 groups
      s0 += __shfl_xor(s0, 16, 64);
      s0 += __shfl_xor(s0, 32, 64);
      if (lane == 0) out[(int64_t)k * Q] = s0;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        float s1 = S1[q][dt];
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (g == 0) out[(int64_t)k * Q + 1 + 16 * dt + li] = s1;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            out[(int64_t)k * Q + 1 + DP + (16 * dt + 4 * g + r) * DP + 16 * b + li] = S2[q][dt][b][r];
      }
    }
  }
  __shared__ float lpw[TR];
  if (tid < TR) lpw[tid] = logp_acc;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int i = 0; i < TR; ++i) s += lpw[i];
    out[(int64_t)K * Q] = s;
  }
}

// one block of DP threads per (chain, component); thread j owns column j / entry j
__global__ void fullcov_finalize(const float* __restrict__ partials, int nb, int64_t n, int D, int K,
                                 const float* __restrict__ theta, const float* __restrict__ Ainv, int C,
                                 float* __restrict__ logp, float* __restrict__ grad) {
  constexpr int Q = 1 + DP + DP * DP;
  const int c = blockIdx.x / K, k = blockIdx.x % K, j = threadIdx.x;
  const int P = K + K * D + K * D * D;
  const int64_t stride = (int64_t)K * Q + 1;
  __shared__ double S2[DP][DP + 1], S1[DP], S0s, A[DP][DP + 1];
  double s1 = 0.0, s0 = 0.0, lp = 0.0;
  double col[DP];
  for (int i = 0; i < DP; ++i) col[i] = 0.0;
  for (int b = 0; b < nb; ++b) {
    const float* p = partials + ((int64_t)b * C + c) * stride + (int64_t)k * Q;
    s1 += (double)p[1 + j];
    for (int i = 0; i < DP; ++i) col[i] += (double)p[1 + DP + i * DP + j];
    if (j == 0) {
      s0 += (double)p[0];
      if (k == 0) lp += (double)partials[((int64_t)b * C + c) * stride + (int64_t)K * Q];
    }
  }
  for (int i = 0; i < DP; ++i) {
    S2[i][j] = col[i];
    A[i][j] = (double)Ainv[((int64_t)c * K + k) * DP * DP + i * DP + j];
  }
  S1[j] = s1;
  if (j == 0) S0s = s0;
  __syncthreads();
  if (j == 0 && k == 0) logp[c] = (float)lp;
  if (!grad) return;
  const float* th = theta + (int64_t)c * P;
  float* gc = grad + (int64_t)c * P;
  if (j == 0) {
    double wsum = 0.0;
    for (int m = 0; m < K; ++m) wsum += (double)th[m];
    gc[k] = (float)(S0s / (double)th[k] - (double)n / wsum);
  }
  if (j < D) {
    // d/dmu_j = sum_m A[m][j] S1[m];  d/dL[i][j] (i >= j) = sum_m A[m][i] S2[m][j] - S0 [i == j] / L_jj
    double gm = 0.0;
    for (int m = 0; m < D; ++m) gm += A[m][j] * S1[m];
    gc[K + k * D + j] = (float)gm;
    const float* L = th + K + K * D + k * D * D;
    for (int i = 0; i < D; ++i) {
      double v = 0.0;
      if (i >= j) {
        for (int m = 0; m < D; ++m) v += A[m][i] * S2[m][j];
        if (i == j) v -= S0s / (double)L[j * D + j];
      }
      gc[K + K * D + k * D * D + i * D + j] = (float)v;
    }
  }
}

int fullcov_blocks(int64_t n, int c) {
  int nb = bvi_ceil_div(n, 4 * TR);
  const int cap = c >= 512 ? 1 : 512 / c;   // ~2 workgroups per CU over all chains
  if (nb > cap) nb = cap;
  return nb < 1 ? 1 : nb;
}

}  // namespace

size_t bvi_ws_gmm_fullcov(int64_t n, int c, int d, int k) {
  (void)d;
  const size_t prep = (size_t)c * k * (DP * DP + DP + 1);
  const size_t part = (size_t)fullcov_blocks(n, c) * c * ((size_t)k * (1 + DP + DP * DP) + 1);
  return (prep + part) * sizeof(float);
}

int bvi_launch_gmm_fullcov(const float* Y, int64_t n, int d, int k, const float* theta, int c, float* logp,
                           float* grad, float* ws, hipStream_t s) {
  if (d < 1 || d > DP || k < 1 || k > 16) return BVI_ERR_UNSUPPORTED;
  float* Ainv = ws;
  float* bvec = Ainv + (size_t)c * k * DP * DP;
  float* cst = bvec + (size_t)c * k * DP;
  float* partials = cst + (size_t)c * k;
  hipLaunchKernelGGL(fullcov_prepare, dim3(c * k), dim3(DP), 0, s, theta, c, k, d, Ainv, bvec, cst);
  BVI_HIP_CHECK_LAUNCH();
  const int nb = fullcov_blocks(n, c);
  const int kq = bvi_ceil_div(k, 4);
  const size_t lds = ((size_t)k * DP * LS + (size_t)k * DP + TR * LS + 4 * kq * TR + TR + 4 * kq) * sizeof(float);
  dim3 grid(nb, c);
#define BVI_FC_CASE(KQ)                                                                                      \
  case KQ: {                                                                                                 \
    hipError_t e = hipFuncSetAttribute((const void*)fullcov_kernel<KQ>,                                      \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
    if (e != hipSuccess) return (int)e;                                                                      \
    hipLaunchKernelGGL(fullcov_kernel<KQ>, grid, dim3(256), lds, s, Y, n, d, k, Ainv, bvec, cst, c, partials); \
  } break;
  switch (kq) {
    BVI_FC_CASE(1)
    BVI_FC_CASE(2)
    BVI_FC_CASE(3)
    BVI_FC_CASE(4)
  }
#undef BVI_FC_CASE
  BVI_HIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(fullcov_finalize, dim3(c * k), dim3(DP), 0, s, (const float*)partials, nb, n, d, k, theta,
                     (const float*)Ainv, c, logp, grad);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}
