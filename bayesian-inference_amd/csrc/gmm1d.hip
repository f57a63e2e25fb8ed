// Univariate Gaussian mixture likelihood and gradient for every chain, streamed
// over y once per chain group with the K-way log-sum-exp kept in registers.
//
// Replaces the 'y' node MixtureSameFamily(Categorical(probs=weights),
// Normal(loc, scale)) of exploratory_experiments/univariate-gaussian-mixture-
// test.ipynb cell 15 (wrapped in tfd.Sample by bayes_vi/model/model.py:17-21)
// and its autodiff gradient.  Maths, per chain and row (w~ = w / sum w):
//   a_k = log w~_k - log s_k - 1/2 log 2pi - u_k^2/2,  u_k = (y - mu_k)/s_k
//   logp += LSE_k a_k,  rho_k = softmax_k a_k
//   dlogp/dmu_k = sum rho_k u_k / s_k,   dlogp/ds_k = sum rho_k (u_k^2 - 1)/s_k
//   dlogp/dw_k  = sum rho_k / w_k - n / sum_j w_j
// The kernel works in base 2 (v_exp_f32 / v_log_f32 are base-2 instructions):
// u' = u sqrt(log2(e)/2), a'_k = B_k - u'^2, and accumulates
//   L2 = sum log2-sum-exp2,  A0_k = sum rho_k,  A1_k = sum rho_k u'_k,  A2_k = sum rho_k u'_k^2;
// finalize rescales them.  Same lane mapping and reduction tree as normal_iid.hip.
#include "bvi_common.h"

template <int K, int CP>
__global__ __launch_bounds__(256) void gmm1d_kernel(const float* __restrict__ y, int64_t n, int kreal,
                                                    const float* __restrict__ theta, int C,
                                                    float* __restrict__ partials) {
  constexpr int RS = 64 / CP;
  constexpr int Q = 1 + 3 * K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cl = lane & (CP - 1), rsub = lane / CP;
  const int chain = blockIdx.y * CP + cl;
  const bool cvalid = chain < C;
  float B[K], mu[K], isg[K];
  {
    const float* th = theta + (int64_t)(cvalid ? chain : 0) * 3 * kreal;
    float wsum = 0.f;
    for (int k = 0; k < kreal; ++k) wsum += th[k];
    const float lw = __log2f(wsum);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k < kreal) {
        const float w = th[k], sc = th[2 * kreal + k];
        mu[k] = th[kreal + k];
        B[k] = __log2f(w) - lw - __log2f(sc) - 0.5f * BVI_LN2PI * BVI_LOG2E;
        isg[k] = 0.84932180028801904f / sc;  // sqrt(log2(e)/2)
      } else {
        mu[k] = 0.f;
        B[k] = -INFINITY;
        isg[k] = 0.f;
      }
    }
  }
  float tot[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) tot[q] = 0.f;

  const int64_t ntiles = (n + 63) / 64;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
    const int64_t i = t * 64 + lane;
    const float yv = i < n ? y[i] : 0.f;
    const bool full = t * 64 + 64 <= n;
    float acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = 0.f;
#pragma unroll 4
    for (int j = 0; j < CP; ++j) {
      const int src = rsub + RS * j;
      const float yj = __shfl(yv, src, 64);
      const float keep = (full || t * 64 + src < n) ? 1.f : 0.f;
      float u[K], a[K];
      float m = -INFINITY;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        u[k] = (yj - mu[k]) * isg[k];
        a[k] = fmaf(-u[k], u[k], B[k]);
        m = fmaxf(m, a[k]);
      }
      float ssum = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        a[k] = __builtin_amdgcn_exp2f(a[k] - m);
        ssum += a[k];
      }
      acc[0] += keep * (m + __builtin_amdgcn_logf(ssum));
      const float inv = keep * __builtin_amdgcn_rcpf(ssum);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float rho = a[k] * inv;
        const float ru = rho * u[k];
        acc[1 + k] += rho;
        acc[1 + K + k] += ru;
        acc[1 + 2 * K + k] = fmaf(ru, u[k], acc[1 + 2 * K + k]);
      }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) tot[q] += acc[q];
  }

  __shared__ float sm[4][CP][Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const float v = bvi_reduce_over_rowsubs<CP>(tot[q]);
    if (rsub == 0) sm[wave][cl][q] = v;
  }
  __syncthreads();
  // raw layout per chain: [L2, A0[kreal], A1[kreal], A2[kreal]]
  const int qreal = 1 + 3 * kreal;
  for (int idx = threadIdx.x; idx < CP * qreal; idx += 256) {
    const int c = idx / qreal, q = idx - c * qreal;
    const int gc = blockIdx.y * CP + c;
    if (gc >= C) continue;
    int qs = 0;
    if (q > 0) {
      const int part = (q - 1) / kreal, k = (q - 1) - part * kreal;
      qs = 1 + part * K + k;
    }
    const float v = sm[0][c][qs] + sm[1][c][qs] + sm[2][c][qs] + sm[3][c][qs];
    partials[((int64_t)blockIdx.x * C + gc) * qreal + q] = v;
  }
}

// Few chains per group (CG = 1 or 4): every lane streams rows of its own with 16-byte loads and carries the
// sums of all CG chains in registers; the chains' parameters are wave-uniform (scalar registers).  The shuffle
// mapping above issues one 4-byte load per lane and iteration, which leaves too little in flight when the
// per-row work is small (one or a few chains).
template <int K, int CG>
__global__ __launch_bounds__(256) void gmm1d_rows_kernel(const float* __restrict__ y, int64_t n, int kreal,
                                                         const float* __restrict__ theta, int C,
                                                         float* __restrict__ partials) {
  constexpr int Q = 1 + 3 * K;
  const int c0 = blockIdx.y * CG;
  float B[CG][K], mu[CG][K], isg[CG][K];
#pragma unroll
  for (int g = 0; g < CG; ++g) {
    const float* th = theta + (int64_t)(c0 + g < C ? c0 + g : 0) * 3 * kreal;   // padding chains: results dropped
    float wsum = 0.f;
    for (int k = 0; k < kreal; ++k) wsum += th[k];
    const float lw = __log2f(wsum);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k < kreal) {
        const float w = th[k], sc = th[2 * kreal + k];
        mu[g][k] = th[kreal + k];
        B[g][k] = __log2f(w) - lw - __log2f(sc) - 0.5f * BVI_LN2PI * BVI_LOG2E;
        isg[g][k] = 0.84932180028801904f / sc;  // sqrt(log2(e)/2)
      } else {
        mu[g][k] = 0.f;
        B[g][k] = -INFINITY;
        isg[g][k] = 0.f;
      }
    }
  }
  float acc[CG][Q];
#pragma unroll
  for (int g = 0; g < CG; ++g)
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[g][q] = 0.f;
  auto row = [&](float yj) {
#pragma unroll
    for (int g = 0; g < CG; ++g) {
      float u[K], a[K];
      float m = -INFINITY;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        u[k] = (yj - mu[g][k]) * isg[g][k];
        a[k] = fmaf(-u[k], u[k], B[g][k]);
        m = fmaxf(m, a[k]);
      }
      float ssum = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        a[k] = __builtin_amdgcn_exp2f(a[k] - m);
        ssum += a[k];
      }
      acc[g][0] += m + __builtin_amdgcn_logf(ssum);
      const float inv = __builtin_amdgcn_rcpf(ssum);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float rho = a[k] * inv;
        const float ru = rho * u[k];
        acc[g][1 + k] += rho;
        acc[g][1 + K + k] += ru;
        acc[g][1 + 2 * K + k] = fmaf(ru, u[k], acc[g][1 + 2 * K + k]);
      }
    }
  };
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  int64_t head = (int64_t)(((16 - ((uintptr_t)y & 15)) & 15) >> 2);
  if (head > n) head = n;
  const f32x4* __restrict__ y4 = reinterpret_cast<const f32x4*>(y + head);
  const int64_t n4 = (n - head) >> 2, tail0 = head + 4 * n4;
  int64_t i = tid;
  for (; i + nth < n4; i += 2 * nth) {
    const f32x4 v0 = y4[i], v1 = y4[i + nth];
    row(v0.x); row(v0.y); row(v0.z); row(v0.w);
    row(v1.x); row(v1.y); row(v1.z); row(v1.w);
  }
  for (; i < n4; i += nth) {
    const f32x4 v = y4[i];
    row(v.x); row(v.y); row(v.z); row(v.w);
  }
  if (tid < head) row(y[tid]);
  if (tid < n - tail0) row(y[tail0 + tid]);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float sm[4][CG][Q];
#pragma unroll
  for (int g = 0; g < CG; ++g)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float v = bvi_reduce_over_rowsubs<1>(acc[g][q]);
      if (lane == 0) sm[wave][g][q] = v;
    }
  __syncthreads();
  // raw layout per chain: [L2, A0[kreal], A1[kreal], A2[kreal]]
  const int qreal = 1 + 3 * kreal;
  for (int idx = threadIdx.x; idx < CG * qreal; idx += 256) {
    const int g = idx / qreal, q = idx - g * qreal;
    if (c0 + g >= C) continue;
    int qs = 0;
    if (q > 0) {
      const int part = (q - 1) / kreal, k = (q - 1) - part * kreal;
      qs = 1 + part * K + k;
    }
    partials[((int64_t)blockIdx.x * C + c0 + g) * qreal + q] =
        (sm[0][g][qs] + sm[1][g][qs]) + (sm[2][g][qs] + sm[3][g][qs]);
  }
}

// one thread (WAVE = false) or one wave (WAVE = true) per gradient entry
template <bool WAVE>
__global__ void gmm1d_finalize(const float* __restrict__ partials, int nb, int64_t n, int k,
                               const float* __restrict__ theta, int C, float* __restrict__ logp,
                               float* __restrict__ grad) {
  const int idx = WAVE ? (blockIdx.x * blockDim.x + threadIdx.x) >> 6 : blockIdx.x * blockDim.x + threadIdx.x;
  const int q3 = 3 * k, qreal = 1 + 3 * k;
  if (idx >= C * q3) return;
  const int c = idx / q3, j = idx - c * q3;  // j indexes the gradient entry (w | mu | scale)
  const int part = j / k, kk = j - part * k;
  const float* th = theta + (int64_t)c * q3;
  const float* pc = partials + (int64_t)c * qreal;
  const int64_t stride = (int64_t)C * qreal;
  const double a0 = bvi_sum_partials<WAVE>(pc + 1 + kk, nb, stride);
  double ax = 0.0, l2 = 0.0;
  if (part == 1) ax = bvi_sum_partials<WAVE>(pc + 1 + k + kk, nb, stride);
  if (part == 2) ax = bvi_sum_partials<WAVE>(pc + 1 + 2 * k + kk, nb, stride);
  if (j == 0) l2 = bvi_sum_partials<WAVE>(pc, nb, stride);
  if (WAVE && (threadIdx.x & 63) != 0) return;
  if (j == 0) logp[c] = (float)(l2 * 0.6931471805599453);
  if (!grad) return;
  const double sc = (double)th[2 * k + kk];
  double g;
  if (part == 0) {
    double wsum = 0.0;
    for (int i = 0; i < k; ++i) wsum += (double)th[i];
    g = a0 / (double)th[kk] - (double)n / wsum;
  } else if (part == 1) {
    g = ax * 1.1774100225154747 / sc;          // sqrt(2 ln 2): u = u' sqrt(2 ln 2)
  } else {
    g = (ax * 1.3862943611198906 - a0) / sc;   // 2 ln 2
  }
  grad[idx] = (float)g;
}

static int gmm1d_blocks(int64_t n) {
  int nb = bvi_ceil_div(n, 256);
  return nb < 1 ? 1 : (nb > BVI_MAX_ROW_BLOCKS ? BVI_MAX_ROW_BLOCKS : nb);
}

size_t bvi_ws_gmm1d(int64_t n, int c, int k) {
  return (size_t)gmm1d_blocks(n) * c * (1 + 3 * k) * sizeof(float);
}

template <int K>
static void gmm1d_dispatch_cp(int cp, dim3 grid, hipStream_t s, const float* y, int64_t n, int k,
                              const float* theta, int c, float* ws) {
  switch (cp) {
    case 1: hipLaunchKernelGGL((gmm1d_rows_kernel<K, 1>), grid, dim3(256), 0, s, y, n, k, theta, c, ws); break;
    case 4:
      if (K <= 8) hipLaunchKernelGGL((gmm1d_rows_kernel<K <= 8 ? K : 8, 4>), grid, dim3(256), 0, s, y, n, k, theta, c, ws);
      else hipLaunchKernelGGL((gmm1d_kernel<K, 4>), grid, dim3(256), 0, s, y, n, k, theta, c, ws);
      break;
    case 16: hipLaunchKernelGGL((gmm1d_kernel<K, 16>), grid, dim3(256), 0, s, y, n, k, theta, c, ws); break;
    default: hipLaunchKernelGGL((gmm1d_kernel<K, 64>), grid, dim3(256), 0, s, y, n, k, theta, c, ws); break;
  }
}

int bvi_launch_gmm1d(const float* y, int64_t n, int k, const float* theta, int c, float* logp, float* grad,
                     float* ws, hipStream_t s) {
  if (k < 1 || k > 16) return BVI_ERR_UNSUPPORTED;
  const int nb = gmm1d_blocks(n);
  const int cp = bvi_chain_pack(c);
  dim3 grid(nb, bvi_ceil_div(c, cp));
  if (k <= 2) gmm1d_dispatch_cp<2>(cp, grid, s, y, n, k, theta, c, ws);
  else if (k <= 4) gmm1d_dispatch_cp<4>(cp, grid, s, y, n, k, theta, c, ws);
  else if (k <= 8) gmm1d_dispatch_cp<8>(cp, grid, s, y, n, k, theta, c, ws);
  else gmm1d_dispatch_cp<16>(cp, grid, s, y, n, k, theta, c, ws);
  BVI_HIP_CHECK_LAUNCH();
  const int tot = c * 3 * k;
  if (nb > 32)
    hipLaunchKernelGGL(gmm1d_finalize<true>, dim3(bvi_ceil_div(tot, 4)), dim3(256), 0, s, (const float*)ws, nb, n, k,
                       theta, c, logp, grad);
  else
    hipLaunchKernelGGL(gmm1d_finalize<false>, dim3(bvi_ceil_div(tot, 128)), dim3(128), 0, s, (const float*)ws, nb, n,
                       k, theta, c, logp, grad);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}
