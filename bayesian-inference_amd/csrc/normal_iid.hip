// iid Normal likelihood: sum_i log N(y_i | loc_c, scale_c) and its gradient for
// every chain c, streamed over y once per chain group.
//
// Replaces the 'y' node of the joint built at bayes_vi/model/model.py:17-21,
// 197-203 (tfd.Sample(tfd.Normal(loc, scale), N), README.md:119-120) and its
// autodiff gradient.  Maths: with r_i = y_i - loc,
//   logp = -S2/(2 scale^2) - n log scale - n/2 log 2pi,  S1 = sum r, S2 = sum r^2
//   dlogp/dloc = S1/scale^2,   dlogp/dscale = -n/scale + S2/scale^3.
//
// Lane mapping: a wave holds CP chains x (64/CP) row subsets, lane = chain +
// CP*rowsub.  Each wave loads 64 consecutive y (one coalesced 256-B read) and
// hands them round with lane shuffles, so every lane sees the rows of its
// subset without re-reading memory.  Sums are two-level in registers (per
// 64-row tile, then running total), butterfly-reduced over row subsets, tree-
// reduced over the 4 waves through LDS, and written as one partial per
// workgroup; finalize adds the partials in fixed order in double.
#include "bvi_common.h"

template <int CP>
__global__ __launch_bounds__(256) void normal_iid_kernel(const float* __restrict__ y, int64_t n,
                                                         const float* __restrict__ theta, int C,
                                                         float* __restrict__ partials) {
  constexpr int RS = 64 / CP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cl = lane & (CP - 1), rsub = lane / CP;
  const int chain = blockIdx.y * CP + cl;
  const float mu = chain < C ? theta[2 * chain] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  const int64_t ntiles = (n + 63) / 64;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
    const int64_t i = t * 64 + lane;
    const float yv = i < n ? y[i] : 0.f;
    float a1 = 0.f, a2 = 0.f;
    if (t * 64 + 64 <= n) {
#pragma unroll
      for (int j = 0; j < CP; ++j) {
        const float r = __shfl(yv, rsub + RS * j, 64) - mu;
        a1 += r;
        a2 = fmaf(r, r, a2);
      }
    } else {
#pragma unroll
      for (int j = 0; j < CP; ++j) {
        const int src = rsub + RS * j;
        float r = __shfl(yv, src, 64) - mu;
        r = (t * 64 + src < n) ? r : 0.f;
        a1 += r;
        a2 = fmaf(r, r, a2);
      }
    }
    s1 += a1;
    s2 += a2;
  }
  s1 = bvi_reduce_over_rowsubs<CP>(s1);
  s2 = bvi_reduce_over_rowsubs<CP>(s2);
  __shared__ float sm[4][CP][2];
  if (rsub == 0) {
    sm[wave][cl][0] = s1;
    sm[wave][cl][1] = s2;
  }
  __syncthreads();
  if (threadIdx.x < CP) {
    const int c = blockIdx.y * CP + threadIdx.x;
    if (c < C) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        t1 += sm[w][threadIdx.x][0];
        t2 += sm[w][threadIdx.x][1];
      }
      float* p = partials + ((int64_t)blockIdx.x * C + c) * 2;
      p[0] = t1;
      p[1] = t2;
    }
  }
}

// one thread (WAVE = false) or one wave (WAVE = true) per chain
template <bool WAVE>
__global__ void normal_iid_finalize(const float* __restrict__ partials, int nb, int64_t n,
                                    const float* __restrict__ theta, int C, float* __restrict__ logp,
                                    float* __restrict__ grad) {
  const int c = WAVE ? (blockIdx.x * blockDim.x + threadIdx.x) >> 6 : blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double s1 = bvi_sum_partials<WAVE>(partials + (int64_t)c * 2, nb, (int64_t)C * 2);
  const double s2 = bvi_sum_partials<WAVE>(partials + (int64_t)c * 2 + 1, nb, (int64_t)C * 2);
  if (WAVE && (threadIdx.x & 63) != 0) return;
  const double sc = (double)theta[2 * c + 1], nn = (double)n;
  const double inv2 = 1.0 / (sc * sc);
  logp[c] = (float)(-0.5 * s2 * inv2 - nn * log(sc) - 0.5 * nn * BVI_LN2PI_D);
  if (grad) {
    grad[2 * c] = (float)(s1 * inv2);
    grad[2 * c + 1] = (float)(-nn / sc + s2 * inv2 / sc);
  }
}

static int normal_iid_blocks(int64_t n) {
  int nb = bvi_ceil_div(n, 256);
  return nb < 1 ? 1 : (nb > BVI_MAX_ROW_BLOCKS ? BVI_MAX_ROW_BLOCKS : nb);
}

size_t bvi_ws_normal_iid(int64_t n, int c) { return (size_t)normal_iid_blocks(n) * c * 2 * sizeof(float); }

int bvi_launch_normal_iid(const float* y, int64_t n, const float* theta, int c, float* logp, float* grad,
                          float* ws, hipStream_t s) {
  const int nb = normal_iid_blocks(n);
  const int cp = bvi_chain_pack(c);
  dim3 grid(nb, bvi_ceil_div(c, cp));
  switch (cp) {
    case 1: hipLaunchKernelGGL(normal_iid_kernel<1>, grid, dim3(256), 0, s, y, n, theta, c, ws); break;
    case 4: hipLaunchKernelGGL(normal_iid_kernel<4>, grid, dim3(256), 0, s, y, n, theta, c, ws); break;
    case 16: hipLaunchKernelGGL(normal_iid_kernel<16>, grid, dim3(256), 0, s, y, n, theta, c, ws); break;
    default: hipLaunchKernelGGL(normal_iid_kernel<64>, grid, dim3(256), 0, s, y, n, theta, c, ws); break;
  }
  BVI_HIP_CHECK_LAUNCH();
  if (nb > 32)
    hipLaunchKernelGGL(normal_iid_finalize<true>, dim3(bvi_ceil_div(c, 4)), dim3(256), 0, s, (const float*)ws, nb, n,
                       theta, c, logp, grad);
  else
    hipLaunchKernelGGL(normal_iid_finalize<false>, dim3(bvi_ceil_div(c, 128)), dim3(128), 0, s, (const float*)ws, nb,
                       n, theta, c, logp, grad);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}
