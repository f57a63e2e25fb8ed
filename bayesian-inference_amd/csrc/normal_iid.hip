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
// subset without re-reading memory; with 64 chains per wave (every lane needs
// every row) the rows are scalar loads instead.  Sums are two-level in registers (per
// 64-row tile, then running total), butterfly-reduced over row subsets, tree-
// reduced over the 4 waves through LDS, and written as one partial per
// workgroup; finalize adds the partials in fixed order in double.
#include "bvi_common.h"
#include <mutex>

// The rows are summed around loc with its 12 lowest mantissa bits cleared: y - loc_hi is then exact for every y
// within 2^12 of loc's size, whereas fl(y - loc) with all 24 bits of loc in play rounds a fixed fraction of the rows
// the same way for some bit patterns of loc (measured at 100M rows: a shift of the log-density by +-0.1 that comes
// and goes with loc's last bits).  normal_iid_finalize moves the sums to loc itself in double:
//   S1 = S1' - n lo,   S2 = S2' - 2 lo S1' + n lo^2,   lo = loc - loc_hi.
__device__ __forceinline__ float normal_loc_hi(float loc) {
  return __uint_as_float(__float_as_uint(loc) & 0xFFFFF000u);
}

template <int CP>
__global__ __launch_bounds__(256) void normal_iid_kernel(const float* __restrict__ y, int64_t n,
                                                         const float* __restrict__ theta, int C,
                                                         float* __restrict__ partials) {
  constexpr int RS = 64 / CP;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform by construction
  const int cl = lane & (CP - 1), rsub = lane / CP;
  const int chain = blockIdx.y * CP + cl;
  const float mu = chain < C ? normal_loc_hi(theta[2 * chain]) : 0.f;
  // S2 ~ n scale^2 is 1e8 times a typical square at 100M rows, and the log-density differences a sampler decides on
  // must come out to a fraction of one square: per-tile sums are floats, their running total is a double, and what
  // leaves the lane is S2 - rows * scale^2 (small near the posterior); finalize adds n * scale^2 back in double
  const double sig2 = chain < C ? (double)theta[2 * chain + 1] * (double)theta[2 * chain + 1] : 0.0;
  float s1 = 0.f;
  double s2 = 0.0;
  int rows = 0;
  double ysum = 0.0;       // lane l: sum of row l of every full tile this wave has taken (CP = 64)
  int rows_full = 0;       // rows of those tiles
  const int64_t ntiles = (n + 63) / 64;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
    float a1 = 0.f, a2 = 0.f;
    if (CP == 64 && t * 64 + 64 <= n) {
      // lane = chain: every lane needs the same 64 rows, so they come through the scalar data path
      // (wave-uniform address, read-only data) and reach the vector ALU as scalar operands -- no LDS
      // round trip per row, two vector instructions per (row, chain): the residual and its square.  The sum of
      // the residuals themselves needs no per-chain work: sum_i (y_i - mu) = sum_i y_i - rows * mu, and sum_i y_i
      // is the same for every chain -- one vector load and one add per tile, row l of the tile in lane l.
      const float* __restrict__ yt = y + t * 64;
      ysum += (double)yt[lane];
      float b2 = 0.f, c2 = 0.f, d2 = 0.f;   // four independent chains of fmas
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float r0 = yt[j] - mu, r1 = yt[16 + j] - mu, r2 = yt[32 + j] - mu, r3 = yt[48 + j] - mu;
        a2 = fmaf(r0, r0, a2); b2 = fmaf(r1, r1, b2); c2 = fmaf(r2, r2, c2); d2 = fmaf(r3, r3, d2);
      }
      s2 += (double)((a2 + b2) + (c2 + d2));
      rows += 64;
      rows_full += 64;
      continue;
    }
    const int64_t i = t * 64 + lane;
    const float yv = i < n ? y[i] : 0.f;
    if (t * 64 + 64 <= n) {
#pragma unroll
      for (int j = 0; j < CP; ++j) {
        const float r = __shfl(yv, rsub + RS * j, 64) - mu;
        a1 += r;
        a2 = fmaf(r, r, a2);
      }
      rows += CP;
    } else {
#pragma unroll
      for (int j = 0; j < CP; ++j) {
        const int src = rsub + RS * j;
        float r = __shfl(yv, src, 64) - mu;
        const bool valid = t * 64 + src < n;
        r = valid ? r : 0.f;
        rows += valid ? 1 : 0;
        a1 += r;
        a2 = fmaf(r, r, a2);
      }
    }
    s1 += a1;
    s2 += (double)a2;
  }
  if (CP == 64) {   // the full tiles' share of S1 (exact in double: mu has 12 mantissa bits, rows_full < 2^31)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ysum += __shfl_xor(ysum, off, 64);
    s1 += (float)(ysum - (double)rows_full * (double)mu);
  }
  s1 = bvi_reduce_over_rowsubs<CP>(s1);
  const float s2c = bvi_reduce_over_rowsubs<CP>((float)(s2 - (double)rows * sig2));
  __shared__ float sm[4][CP][2];
  if (rsub == 0) {
    sm[wave][cl][0] = s1;
    sm[wave][cl][1] = s2c;
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

// Few chains per group (CG = 1, 4, 16): every lane streams rows of its own -- 16-byte loads, four in flight --
// and carries the sums of all CG chains in registers; no lane exchange until the final reduction.
// (With one 4-byte load per lane and iteration, as in the shuffle mapping above, a wave has 256 B in flight
// and the loop runs at memory latency, not bandwidth.)
template <int CG>
__global__ __launch_bounds__(256) void normal_iid_rows_kernel(const float* __restrict__ y, int64_t n,
                                                              const float* __restrict__ theta, int C,
                                                              float* __restrict__ partials) {
  constexpr int U = 4;
  const int c0 = blockIdx.y * CG;
  float mu[CG], s1[CG], s2[CG];
#pragma unroll
  for (int k = 0; k < CG; ++k) {
    mu[k] = c0 + k < C ? normal_loc_hi(theta[2 * (c0 + k)]) : 0.f;
    s1[k] = 0.f;
    s2[k] = 0.f;
  }
  // The residuals are summed for the group's first chain only: sum_i (y_i - mu_k) = sum_i (y_i - mu_0) +
  // rows (mu_0 - mu_k), exact in double (the mu have 12 mantissa bits) -- 2 CG + 1 instead of 3 CG operations per row.
  auto row = [&](float yv) {
#pragma unroll
    for (int k = 0; k < CG; ++k) {
      const float r = yv - mu[k];
      if (k == 0) s1[0] += r;
      s2[k] = fmaf(r, r, s2[k]);
    }
  };
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  // rows before the first 16-byte boundary, then whole float4s, then the last few rows
  int64_t head = (int64_t)(((16 - ((uintptr_t)y & 15)) & 15) >> 2);
  if (head > n) head = n;
  const f32x4* __restrict__ y4 = reinterpret_cast<const f32x4*>(y + head);
  const int64_t n4 = (n - head) >> 2, tail0 = head + 4 * n4;
  int64_t i = tid;
  for (; i + (U - 1) * nth < n4; i += U * nth) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = y4[i + u * nth];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      row(v[u].x); row(v[u].y); row(v[u].z); row(v[u].w);
    }
  }
  for (; i < n4; i += nth) {
    const f32x4 v = y4[i];
    row(v.x); row(v.y); row(v.z); row(v.w);
  }
  if (tid < head) row(y[tid]);
  if (tid < n - tail0) row(y[tail0 + tid]);
  // rows this lane has summed (a few hundred at most: its float sums are accurate); what leaves the lane is
  // S2 - rows * scale^2, small near the posterior, and finalize adds n * scale^2 back in double
  const double rows = (double)(4 * (tid < n4 ? (n4 - tid + nth - 1) / nth : 0) + (tid < head ? 1 : 0) +
                               (tid < n - tail0 ? 1 : 0));

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float sm[4][CG][2];
#pragma unroll
  for (int k = 0; k < CG; ++k) {
    const double sc = c0 + k < C ? (double)theta[2 * (c0 + k) + 1] : 0.0;
    const float t1 = bvi_reduce_over_rowsubs<1>(
        k == 0 ? s1[0] : (float)((double)s1[0] + rows * ((double)mu[0] - (double)mu[k])));
    const float t2 = bvi_reduce_over_rowsubs<1>((float)((double)s2[k] - rows * sc * sc));
    if (lane == 0) {
      sm[wave][k][0] = t1;
      sm[wave][k][1] = t2;
    }
  }
  __syncthreads();
  if (threadIdx.x < CG && c0 + threadIdx.x < C) {
    const int k = threadIdx.x;
    float* p = partials + ((int64_t)blockIdx.x * C + c0 + k) * 2;
    p[0] = (sm[0][k][0] + sm[1][k][0]) + (sm[2][k][0] + sm[3][k][0]);
    p[1] = (sm[0][k][1] + sm[1][k][1]) + (sm[2][k][1] + sm[3][k][1]);
  }
}

// one thread (WAVE = false) or one wave (WAVE = true) per chain
template <bool WAVE>
__global__ void normal_iid_finalize(const float* __restrict__ partials, int nb, int64_t n,
                                    const float* __restrict__ theta, int C, float* __restrict__ logp,
                                    float* __restrict__ grad, const double* __restrict__ ref) {
  const int c = WAVE ? (blockIdx.x * blockDim.x + threadIdx.x) >> 6 : blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double s1h = bvi_sum_partials<WAVE>(partials + (int64_t)c * 2, nb, (int64_t)C * 2);
  const double s2c = bvi_sum_partials<WAVE>(partials + (int64_t)c * 2 + 1, nb, (int64_t)C * 2);
  if (WAVE && (threadIdx.x & 63) != 0) return;
  const double sc = (double)theta[2 * c + 1], nn = (double)n;
  const double inv2 = 1.0 / (sc * sc);
  // the kernels hand over S1', S2' - rows * scale^2 taken around loc_hi
  const double lo = (double)theta[2 * c] - (double)normal_loc_hi(theta[2 * c]);
  const double s1 = s1h - nn * lo;
  const double s2 = s2c + nn * sc * sc - 2.0 * lo * s1h + nn * lo * lo;
  logp[c] = (float)(-0.5 * s2 * inv2 - nn * log(sc) - 0.5 * nn * BVI_LN2PI_D - (ref ? *ref : 0.0));
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
                          float* ws, hipStream_t s, const double* logp_ref) {
  const int nb = normal_iid_blocks(n);
  const int cp = bvi_chain_pack(c);
  dim3 grid(nb, bvi_ceil_div(c, cp));
  switch (cp) {
    case 1: hipLaunchKernelGGL(normal_iid_rows_kernel<1>, grid, dim3(256), 0, s, y, n, theta, c, ws); break;
    case 4: hipLaunchKernelGGL(normal_iid_rows_kernel<4>, grid, dim3(256), 0, s, y, n, theta, c, ws); break;
    case 16: hipLaunchKernelGGL(normal_iid_rows_kernel<16>, grid, dim3(256), 0, s, y, n, theta, c, ws); break;
    default: hipLaunchKernelGGL(normal_iid_kernel<64>, grid, dim3(256), 0, s, y, n, theta, c, ws); break;
  }
  BVI_HIP_CHECK_LAUNCH();
  if (nb > 32)
    hipLaunchKernelGGL(normal_iid_finalize<true>, dim3(bvi_ceil_div(c, 4)), dim3(256), 0, s, (const float*)ws, nb, n,
                       theta, c, logp, grad, logp_ref);
  else
    hipLaunchKernelGGL(normal_iid_finalize<false>, dim3(bvi_ceil_div(c, 128)), dim3(128), 0, s, (const float*)ws, nb,
                       n, theta, c, logp, grad, logp_ref);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

// ---- the allocation-free variant: scratch in a file-scope device array instead of a caller workspace ----
// For callers that cannot allocate device memory (code under HIP graph capture, tight serving loops).  The
// per-row-block partial sums of the kernels above go to `scratch`; every launch writes each partial it later
// reads (all row blocks x all chains of the call), so nothing has to be cleared between calls.  All calls on one
// GPU share that GPU's array: calls that overlap in time on one GPU race.
static __device__ __align__(256) float scratch[(size_t)BVI_MAX_ROW_BLOCKS * BVI_NOALLOC_MAX_CHAINS * 2];

#define BVI_MAX_DEVICES 64
static float* scratch_ptr[BVI_MAX_DEVICES];   // device address of `scratch` on each GPU, looked up once per device
static std::mutex scratch_lock;

// The array's name is only a host-side handle; the kernels need the address hipGetSymbolAddress reports for the
// current device (each GPU holds its own copy at its own address).
static int noalloc_scratch(float** out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= BVI_MAX_DEVICES) return BVI_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> guard(scratch_lock);
  if (!scratch_ptr[dev]) {
    float* ptr = nullptr;
    e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return (int)e;
    if (!ptr) return BVI_ERR_ARG;
    scratch_ptr[dev] = ptr;
  }
  *out = scratch_ptr[dev];
  return BVI_OK;
}

extern "C" int bvi_normal_iid_logp_grad_noalloc(const float* y, int64_t n, const float* theta, int n_chains,
                                                float* logp, float* grad, void* stream) {
  if ((!y && n > 0) || !theta || !logp || n < 0 || n_chains < 1) return BVI_ERR_ARG;
  if (n_chains > BVI_NOALLOC_MAX_CHAINS) return BVI_ERR_TOO_LARGE;
  if (bvi_ws_normal_iid(n, n_chains) > sizeof(float) * (size_t)BVI_MAX_ROW_BLOCKS * BVI_NOALLOC_MAX_CHAINS * 2)
    return BVI_ERR_TOO_LARGE;
  float* ws = nullptr;
  const int st = noalloc_scratch(&ws);
  if (st != BVI_OK) return st;
  return bvi_launch_normal_iid(y, n, theta, n_chains, logp, grad, ws, (hipStream_t)stream);
}
