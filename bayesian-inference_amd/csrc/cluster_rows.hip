// One-time row order for the full-covariance mixture kernels: rows that lie close together are stored next to each
// other.  The rows of a likelihood sum are exchangeable (the reference sums them with tf.reduce_sum inside
// tfd.Sample / MixtureSameFamily.log_prob, bayes_vi/model/model.py:17-21), so their order in memory is the layout's to
// choose; gmm_fullcov.hip leaves a component out of every 64-row tile it cannot reach, which is most tiles once
// neighbouring rows belong to the same few components.
//
// Lloyd's algorithm on m <= 128 centres: start from m rows at a fixed stride, two updates, then a stable counting sort
// of the rows by their centre.  Everything is reproducible bit for bit: nearest centre with ties to the lowest index,
// centre sums in 64-bit fixed point (integer additions commute, so the atomics' order does not matter), stable ranks
// inside the sort.  Rows with a non-finite entry go to centre 0 and take no part in the sums.
#include "bvi_common.h"

namespace {

constexpr int kDP = 32;              // row length in registers (d <= 32, zero padded)
constexpr int kMaxCentres = 128;     // the centre sums of a workgroup (128 x 33 64-bit words) and the centres fit 64 KB of LDS
constexpr int kAssignThreads = 256;
constexpr int kMaxSortBlocks = 2048;

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

struct Layout {
  size_t centres, sums, amax, assign, hist, offs, total;
  int64_t rows_per_block;
  int nblocks;
};

Layout layout(int64_t n, int d, int m) {
  Layout L;
  int64_t r = (n + kMaxSortBlocks - 1) / kMaxSortBlocks;
  r = ((r < 4096 ? 4096 : r) + 63) / 64 * 64;
  L.rows_per_block = r;
  L.nblocks = n > 0 ? (int)((n + r - 1) / r) : 0;
  size_t o = 0;
  L.centres = o; o += align256((size_t)m * d * sizeof(float));
  L.sums = o;    o += align256((size_t)m * (d + 1) * sizeof(long long));
  L.amax = o;    o += 256;
  L.assign = o;  o += align256((size_t)n);
  L.hist = o;    o += align256((size_t)L.nblocks * m * sizeof(unsigned));
  L.offs = o;    o += align256((size_t)L.nblocks * m * sizeof(long long));
  L.total = o;
  return L;
}

// largest finite |entry| as the bits of a non-negative float (integer max: the order of the atomics does not matter)
__global__ __launch_bounds__(256) void cluster_absmax(const float* __restrict__ Y, int64_t total, unsigned* __restrict__ amax) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float a = fabsf(Y[i]);
    if (isfinite(a)) m = fmaxf(m, a);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(amax, __float_as_uint(m));
}

__global__ void cluster_init(const float* __restrict__ Y, int64_t n, int d, int m, float* __restrict__ centres) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m * d) return;
  const int c = e / d, j = e % d;
  const int64_t stride = n / m;                  // m <= n
  const int64_t row = c * stride + stride / 2;
  const float v = Y[row * d + j];
  centres[e] = isfinite(v) ? v : 0.f;
}

// fixed-point scale of the centre sums: |entry| 2^e < 2^(61 - nbits) with n < 2^nbits
__device__ __forceinline__ int fixed_exponent(unsigned amax_bits, int nbits) {
  const float a = __uint_as_float(amax_bits);
  return (a > 0.f ? -ilogbf(a) - 1 : 0) + 61 - nbits;
}

template <bool ACCUM>
__global__ __launch_bounds__(kAssignThreads) void cluster_assign(const float* __restrict__ Y, int64_t n, int d, int m,
                                                                 const float* __restrict__ centres,
                                                                 const unsigned* __restrict__ amax, int nbits,
                                                                 unsigned char* __restrict__ assign,
                                                                 unsigned long long* __restrict__ sums) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* cs = reinterpret_cast<float*>(smem);                          // [m][kDP], zero padded
  float* cn = cs + m * kDP;                                            // [m]
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(cn + ((m + 3) & ~3));   // [m][d + 1]
  for (int e = threadIdx.x; e < m * kDP; e += kAssignThreads) {
    const int c = e / kDP, j = e % kDP;
    cs[e] = j < d ? centres[c * d + j] : 0.f;
  }
  if (ACCUM)
    for (int e = threadIdx.x; e < m * (d + 1); e += kAssignThreads) acc[e] = 0ull;
  __syncthreads();
  for (int c = threadIdx.x; c < m; c += kAssignThreads) {
    float s = 0.f;
    for (int j = 0; j < kDP; ++j) s = fmaf(cs[c * kDP + j], cs[c * kDP + j], s);
    cn[c] = s;
  }
  __syncthreads();
  const double scale = ACCUM ? ldexp(1.0, fixed_exponent(*amax, nbits)) : 0.0;
  for (int64_t row = (int64_t)blockIdx.x * kAssignThreads + threadIdx.x; row < n;
       row += (int64_t)gridDim.x * kAssignThreads) {
    float y[kDP];
    bool finite = true;
#pragma unroll
    for (int j = 0; j < kDP; ++j) {
      y[j] = j < d ? Y[row * d + j] : 0.f;
      finite = finite && isfinite(y[j]);
    }
    int best = 0;
    if (finite) {
      float bs = INFINITY;
      for (int c = 0; c < m; ++c) {
        const f32x4* cp = reinterpret_cast<const f32x4*>(cs + c * kDP);       // every lane reads the same address
        float dot = 0.f;
#pragma unroll
        for (int q = 0; q < kDP / 4; ++q) {
          const f32x4 v = cp[q];
          dot = fmaf(y[4 * q], v[0], dot);
          dot = fmaf(y[4 * q + 1], v[1], dot);
          dot = fmaf(y[4 * q + 2], v[2], dot);
          dot = fmaf(y[4 * q + 3], v[3], dot);
        }
        const float sc = fmaf(-2.f, dot, cn[c]);
        if (sc < bs) { bs = sc; best = c; }
      }
    }
    assign[row] = (unsigned char)best;
    if (ACCUM && finite) {
      unsigned long long* a = acc + best * (d + 1);
#pragma unroll
      for (int j = 0; j < kDP; ++j)
        if (j < d) atomicAdd(a + j, (unsigned long long)__double2ll_rn((double)y[j] * scale));
      atomicAdd(a + d, 1ull);
    }
  }
  if (ACCUM) {
    __syncthreads();
    for (int e = threadIdx.x; e < m * (d + 1); e += kAssignThreads)
      if (acc[e] != 0ull) atomicAdd(sums + e, acc[e]);
  }
}

__global__ void cluster_update(int d, int m, const unsigned long long* __restrict__ sums,
                               const unsigned* __restrict__ amax, int nbits, float* __restrict__ centres) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m * d) return;
  const int c = e / d, j = e % d;
  const long long cnt = (long long)sums[c * (d + 1) + d];
  if (cnt > 0)     // a centre that lost all its rows stays where it is
    centres[e] = (float)(ldexp((double)(long long)sums[c * (d + 1) + j], -fixed_exponent(*amax, nbits)) / (double)cnt);
}

// ---- stable counting sort by centre: rows of sort block b are [b R, (b + 1) R) ----------------------------------
__global__ __launch_bounds__(256) void cluster_hist(const unsigned char* __restrict__ assign, int64_t n, int64_t R,
                                                    int m, unsigned* __restrict__ hist) {
  __shared__ unsigned h[kMaxCentres];
  for (int c = threadIdx.x; c < m; c += 256) h[c] = 0u;
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * R, hi = lo + R < n ? lo + R : n;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&h[assign[i]], 1u);
  __syncthreads();
  for (int c = threadIdx.x; c < m; c += 256) hist[(int64_t)blockIdx.x * m + c] = h[c];
}

// offs[b][c] = rows of centres < c + rows of centre c in sort blocks < b; one workgroup, thread = centre
__global__ __launch_bounds__(kMaxCentres) void cluster_scan(const unsigned* __restrict__ hist, int nblocks, int m,
                                                            long long* __restrict__ offs) {
  __shared__ long long tot[kMaxCentres];
  const int c = threadIdx.x;
  long long run = 0;
  if (c < m)
    for (int b = 0; b < nblocks; ++b) {
      offs[(int64_t)b * m + c] = run;
      run += hist[(int64_t)b * m + c];
    }
  tot[c] = c < m ? run : 0;
  __syncthreads();
  long long base = 0;
  for (int q = 0; q < c; ++q) base += tot[q];
  if (c < m)
    for (int b = 0; b < nblocks; ++b) offs[(int64_t)b * m + c] += base;
}

// one wave per sort block, 64 rows at a time in row order: the rows of one centre among the 64 take consecutive
// places in lane order.  The running place of centre c sits in lane c % 64, register c / 64.
__global__ __launch_bounds__(64) void cluster_scatter(const float* __restrict__ Y, int64_t n, int d, int64_t R, int m,
                                                      const unsigned char* __restrict__ assign,
                                                      const long long* __restrict__ offs, float* __restrict__ Yout,
                                                      int64_t* __restrict__ order) {
  const int lane = threadIdx.x;
  long long place[kMaxCentres / 64];
#pragma unroll
  for (int q = 0; q < kMaxCentres / 64; ++q)
    place[q] = 64 * q + lane < m ? offs[(int64_t)blockIdx.x * m + 64 * q + lane] : 0;
  const int64_t lo = (int64_t)blockIdx.x * R, hi = lo + R < n ? lo + R : n;
  for (int64_t base = lo; base < hi; base += 64) {
    const int64_t row = base + lane;
    const bool valid = row < hi;
    const int key = valid ? (int)assign[row] : -1;
    long long pos = 0;
    unsigned long long todo = __ballot(valid);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int kk = __shfl(key, leader, 64);
      const unsigned long long mask = __ballot(key == kk);
      long long cur = 0;
#pragma unroll
      for (int q = 0; q < kMaxCentres / 64; ++q)
        if ((kk >> 6) == q) cur = __shfl(place[q], kk & 63, 64);
      if (key == kk) pos = cur + __popcll(mask & ((1ull << lane) - 1ull));
      const int cnt = __popcll(mask);
#pragma unroll
      for (int q = 0; q < kMaxCentres / 64; ++q)
        if ((kk >> 6) == q && lane == (kk & 63)) place[q] += cnt;
      todo &= ~mask;
    }
    if (valid) {
      for (int j = 0; j < d; ++j) Yout[pos * d + j] = Y[row * d + j];
      if (order) order[pos] = row;
    }
  }
}

}  // namespace

size_t bvi_ws_cluster_rows(int64_t n, int d, int m) { return layout(n, d, m).total; }

int bvi_launch_cluster_rows(const float* Y, int64_t n, int d, int m, float* Yout, int64_t* order, void* ws,
                            hipStream_t s) {
  if (n == 0) return BVI_OK;
  if (m > n) m = (int)n;
  const Layout L = layout(n, d, m);
  unsigned char* base = static_cast<unsigned char*>(ws);
  float* centres = reinterpret_cast<float*>(base + L.centres);
  unsigned long long* sums = reinterpret_cast<unsigned long long*>(base + L.sums);
  unsigned* amax = reinterpret_cast<unsigned*>(base + L.amax);
  unsigned char* assign = base + L.assign;
  unsigned* hist = reinterpret_cast<unsigned*>(base + L.hist);
  long long* offs = reinterpret_cast<long long*>(base + L.offs);
  int nbits = 1;
  while (((int64_t)1 << nbits) <= n) ++nbits;
  const int64_t total = n * d;
  const int grid = (int)((n + kAssignThreads - 1) / kAssignThreads < 1024 ? (n + kAssignThreads - 1) / kAssignThreads
                                                                          : 1024);
  const size_t lds_assign = (size_t)m * kDP * sizeof(float) + (size_t)((m + 3) & ~3) * sizeof(float);
  const size_t lds_accum = lds_assign + (size_t)m * (d + 1) * sizeof(unsigned long long);
  hipError_t e = hipMemsetAsync(amax, 0, sizeof(unsigned), s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(cluster_absmax, dim3(1024), dim3(256), 0, s, Y, total, amax);
  hipLaunchKernelGGL(cluster_init, dim3((m * d + 255) / 256), dim3(256), 0, s, Y, n, d, m, centres);
  for (int it = 0; it < 2; ++it) {
    e = hipMemsetAsync(sums, 0, (size_t)m * (d + 1) * sizeof(unsigned long long), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cluster_assign<true>, dim3(grid), dim3(kAssignThreads), lds_accum, s, Y, n, d, m,
                       (const float*)centres, (const unsigned*)amax, nbits, assign, sums);
    hipLaunchKernelGGL(cluster_update, dim3((m * d + 255) / 256), dim3(256), 0, s, d, m,
                       (const unsigned long long*)sums, (const unsigned*)amax, nbits, centres);
  }
  hipLaunchKernelGGL(cluster_assign<false>, dim3(grid), dim3(kAssignThreads), lds_assign, s, Y, n, d, m,
                     (const float*)centres, (const unsigned*)amax, nbits, assign, (unsigned long long*)nullptr);
  hipLaunchKernelGGL(cluster_hist, dim3(L.nblocks), dim3(256), 0, s, (const unsigned char*)assign, n,
                     L.rows_per_block, m, hist);
  hipLaunchKernelGGL(cluster_scan, dim3(1), dim3(kMaxCentres), 0, s, (const unsigned*)hist, L.nblocks, m, offs);
  hipLaunchKernelGGL(cluster_scatter, dim3(L.nblocks), dim3(64), 0, s, Y, n, d, L.rows_per_block, m,
                     (const unsigned char*)assign, (const long long*)offs, Yout, order);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}
