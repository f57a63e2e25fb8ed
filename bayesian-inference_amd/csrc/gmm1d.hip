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

// Per chain: component means, scaled inverse widths and log2 heights relative to the tallest component
// (hmax = max_k B_k, so every exponent below is <= 0 and no per-row maximum is needed).
template <int K>
struct MixParams {
  float mu[K], isg[K], Bp[K];
  float hmax;
};

template <int K>
__device__ __forceinline__ void mix_load_params(const float* __restrict__ th, int kreal, MixParams<K>& p) {
  float wsum = 0.f;
  for (int k = 0; k < kreal; ++k) wsum += th[k];
  const float lw = __log2f(wsum);
  float B[K];
  p.hmax = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (k < kreal) {
      const float w = th[k], sc = th[2 * kreal + k];
      p.mu[k] = th[kreal + k];
      B[k] = __log2f(w) - lw - __log2f(sc) - 0.5f * BVI_LN2PI * BVI_LOG2E;
      p.isg[k] = 0.84932180028801904f / sc;  // sqrt(log2(e)/2)
      p.hmax = fmaxf(p.hmax, B[k]);
    } else {
      p.mu[k] = 0.f;
      B[k] = -INFINITY;
      p.isg[k] = 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) p.Bp[k] = B[k] - p.hmax;
}

// One row with the per-row maximum subtracted before the exponentials: the form that cannot underflow,
// taken only by rows that lie far from every component (mix_row sends them here).
template <int K, bool GRAD>
__device__ __forceinline__ void mix_row_far(float yj, float keep, const MixParams<K>& p, float (&acc)[1 + 3 * K]) {
  float u[K], a[K];
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    u[k] = (yj - p.mu[k]) * p.isg[k];
    a[k] = fmaf(-u[k], u[k], p.Bp[k]);
    m = fmaxf(m, a[k]);
  }
  float ssum = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    a[k] = __builtin_amdgcn_exp2f(a[k] - m);
    ssum += a[k];
  }
  acc[0] += keep * (m + __builtin_amdgcn_logf(ssum));
  if (GRAD) {
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
}

#ifndef BVI_GMM_NEAR_LOG2
#define BVI_GMM_NEAR_LOG2 8
#endif
#ifndef BVI_GMM_SKIP_LOG2
#define BVI_GMM_SKIP_LOG2 40
#endif
static constexpr float kMixNear = 1.f / (float)(1ull << BVI_GMM_NEAR_LOG2);
static constexpr float kMixSkip = (float)BVI_GMM_SKIP_LOG2;
static constexpr int kMixMaxBins = 512;
// What a lane hands on is sum_i (v_i - kMixRowLevel), v_i = log2 sum_k 2^(Bp_k - u'_ik^2): a row at a typical distance
// from its component has v ~ -log2(e)/2, so the centred sums stay random-walk sized (floats hold them to ~1e-5 at
// 100M rows, where the uncentred total is ~1e8); gmm1d_finalize adds n * kMixRowLevel back in double.
static constexpr float kMixRowLevel = -0.72134752f;
#ifndef BVI_GMM_UNROLL
#define BVI_GMM_UNROLL 1
#endif
#define BVI_PRAGMA_(x) _Pragma(#x)
#define BVI_PRAGMA(x) BVI_PRAGMA_(x)

// One row for the chain(s) of this lane.  Exponents are taken relative to the tallest component, so a row costs
// (sub, mul, fma, exp2, add) per component and one log2; rows whose sum falls below 2^-8 of the tallest peak
// (about 3.3 widths from the nearest component) are redone by mix_row_far.  With SKIP, only the components whose
// bit is set in `act` (wave-uniform, see MixTable) are evaluated at all: a component left out has an exponent
// below -40 in every lane of the wave, so what it would add is below 2^-32 of the sum that is kept.
template <int K, bool GRAD, bool SKIP>
__device__ __forceinline__ void mix_row(float yj, float keep, unsigned act, const MixParams<K>& p,
                                        float (&acc)[1 + 3 * K]) {
  float u[K], e[K];
  if (SKIP) {   // u[k], e[k] are read only where they were set; an opaque definition spares the merges a copy each
#pragma unroll
    for (int k = 0; k < K; ++k) {
      asm volatile("" : "=v"(e[k]));
      asm volatile("" : "=v"(u[k]));
    }
  }
  float ssum = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (!SKIP || (act & (1u << k))) {
      if (SKIP) asm volatile("" ::);   // keeps the branch a branch: the body is not worth speculating
      u[k] = (yj - p.mu[k]) * p.isg[k];
      e[k] = __builtin_amdgcn_exp2f(fmaf(-u[k], u[k], p.Bp[k]));
      ssum += e[k];
    }
  }
  const bool near = ssum >= kMixNear;   // false for NaN as well: mix_row_far then produces the NaN
  acc[0] += near ? keep * __builtin_amdgcn_logf(ssum) : 0.f;
  if (GRAD) {
    const float inv = near ? keep * __builtin_amdgcn_rcpf(ssum) : 0.f;
    // a copy the compiler cannot relate to `act`: it then tests one bit per component here as above instead of
    // carrying eight 64-bit lane masks from the first pass to this one
    unsigned act2 = act;
    if (SKIP) asm volatile("" : "+s"(act2));
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (!SKIP || (act2 & (1u << k))) {
        if (SKIP) asm volatile("" ::);
        const float rho = e[k] * inv;
        const float ru = rho * u[k];
        acc[1 + k] += rho;
        acc[1 + K + k] += ru;
        acc[1 + 2 * K + k] = fmaf(ru, u[k], acc[1 + 2 * K + k]);
      }
    }
  }
  if (__builtin_expect(!near, 0)) mix_row_far<K, GRAD>(yj, keep, p, acc);
}

// Four rows per lane with one test per component for all four (the few-chain kernel, whose lanes hold rows: the
// scalar test-and-branch per component costs as much as a component's arithmetic for one row).
template <int K, bool GRAD, int R = 4>
__device__ __forceinline__ void mix_rows4_skip(const float (&y)[R], unsigned act, const MixParams<K>& p,
                                               float (&acc)[1 + 3 * K]) {
  float u[GRAD ? K : 1][R], e[GRAD ? K : 1][R];
  if (GRAD) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        asm volatile("" : "=v"(e[k][r]));
        asm volatile("" : "=v"(u[k][r]));
      }
  }
  float ssum[R];
#pragma unroll
  for (int r = 0; r < R; ++r) ssum[r] = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (act & (1u << k)) {
      asm volatile("" ::);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float uu = (y[r] - p.mu[k]) * p.isg[k];
        const float ee = __builtin_amdgcn_exp2f(fmaf(-uu, uu, p.Bp[k]));
        ssum[r] += ee;
        if (GRAD) { u[k][r] = uu; e[k][r] = ee; }
      }
    }
  }
  bool near[R];
  float inv[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    near[r] = ssum[r] >= kMixNear;
    acc[0] += near[r] ? __builtin_amdgcn_logf(ssum[r]) : 0.f;
    if (GRAD) inv[r] = near[r] ? __builtin_amdgcn_rcpf(ssum[r]) : 0.f;
  }
  if (GRAD) {
    unsigned act2 = act;
    asm volatile("" : "+s"(act2));
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (act2 & (1u << k)) {
        asm volatile("" ::);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float rho = e[k][r] * inv[r];
          const float ru = rho * u[k][r];
          acc[1 + k] += rho;
          acc[1 + K + k] += ru;
          acc[1 + 2 * K + k] = fmaf(ru, u[k][r], acc[1 + 2 * K + k]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (__builtin_expect(!near[r], 0)) mix_row_far<K, GRAD>(y[r], 1.f, p, acc);
}

// Which components can matter at a given y, for the chains of one wave: the y axis between the outermost reaches
// of the wave's components is cut into `nbins` bins, and tab[1 + b] holds a bit per component that is set unless,
// for every chain of the wave, the whole bin (widened by a bin on each side, which covers the rounding of the bin
// index) lies where the component's exponent is below -kMixSkip.  tab[0] and tab[nbins + 1] (y outside every
// reach) are 0: such rows take mix_row_far.  Non-finite parameters compare false and leave the bit set.
template <int K>
struct MixTable {
  float y0, scale;
  int nbins;
  __device__ __forceinline__ void build(const MixParams<K>& p, int kreal, int nb, unsigned short* tab) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    nbins = nb;
    float lo[K], hi[K];
    float ymin = INFINITY, ymax = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k < kreal) {
        const float reach = sqrtf(fmaxf(kMixSkip + p.Bp[k], 0.f)) / p.isg[k] * 1.0001f;
        lo[k] = p.mu[k] - reach;
        hi[k] = p.mu[k] + reach;
        ymin = fminf(ymin, lo[k]);
        ymax = fmaxf(ymax, hi[k]);
      } else {
        lo[k] = INFINITY;
        hi[k] = -INFINITY;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      ymin = fminf(ymin, __shfl_xor(ymin, off, 64));
      ymax = fmaxf(ymax, __shfl_xor(ymax, off, 64));
    }
    y0 = ymin;
    const float width = fmaxf(ymax - ymin, 1e-30f) / (float)nb;
    scale = 1.f / width;
    for (int b = wave; b < nb; b += 4) {
      const float blo = y0 + (float)(b - 1) * width, bhi = y0 + (float)(b + 2) * width;
      unsigned m = 0;
#pragma unroll
      for (int k = 0; k < K; ++k)
        m |= (__builtin_amdgcn_ballot_w64(!(hi[k] < blo || lo[k] > bhi)) != 0 ? 1u : 0u) << k;
      if (lane == 0) tab[1 + b] = (unsigned short)m;
    }
    if (threadIdx.x == 0) tab[0] = tab[nb + 1] = 0u;
  }
  // table entry for the row each lane holds (NaN rows read entry 0 and come out NaN through mix_row_far)
  __device__ __forceinline__ unsigned lookup(float yv, const unsigned short* tab) const {
    const float fb = fminf(fmaxf(floorf((yv - y0) * scale), -1.f), (float)nbins);
    return tab[(int)fb + 1];
  }
};

// raw layout per chain in the partials: [L2, A0[kreal], A1[kreal], A2[kreal]]; a value-only launch writes L2 only.
// red[q][64 * wave + c] holds wave `wave`'s sum q for the c-th chain of the block.
template <int K, int NC, bool GRAD>
__device__ __forceinline__ void mix_store_partials(const float (*red)[256], int kreal, int c0, int C,
                                                   float* __restrict__ partials) {
  const int qreal = 1 + 3 * kreal, qw = GRAD ? qreal : 1;
  for (int idx = threadIdx.x; idx < NC * qw; idx += 256) {
    const int c = idx / qw, q = idx - c * qw;
    if (c0 + c >= C) continue;
    int qs = 0;
    if (q > 0) {
      const int part = (q - 1) / kreal, k = (q - 1) - part * kreal;
      qs = 1 + part * K + k;
    }
    partials[((int64_t)blockIdx.x * C + c0 + c) * qreal + q] =
        (red[qs][c] + red[qs][64 + c]) + (red[qs][128 + c] + red[qs][192 + c]);
  }
}

#ifndef BVI_GMM_WAVES
#define BVI_GMM_WAVES 6
#endif
template <int K, int CP, bool GRAD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(K <= 8 ? (CP == 64 ? BVI_GMM_WAVES : 4) : 3)))   // fewer chains per wave: no spills at 4
void gmm1d_kernel(const float* __restrict__ y, int64_t n, int kreal,
                                                    const float* __restrict__ theta, int C, int nbins,
                                                    float* __restrict__ partials) {
  constexpr int RS = 64 / CP;
  constexpr int Q = 1 + 3 * K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cl = lane & (CP - 1), rsub = lane / CP;
  const int chain = blockIdx.y * CP + cl;
  MixParams<K> p;
  mix_load_params<K>(theta + (int64_t)(chain < C ? chain : 0) * 3 * kreal, kreal, p);
  __shared__ unsigned short tab[kMixMaxBins + 2];
  MixTable<K> table;
  if (nbins > 0) {   // all four waves hold the same chains, so one table serves the block
    table.build(p, kreal, nbins, tab);
    __syncthreads();
  }
  // Sums over this lane's rows are kept at two levels (per tile in registers, per kernel below) so that no float
  // accumulates more than a few hundred addends; the kernel-level gradient sums live in LDS, one column per
  // thread, which leaves registers for six waves per SIMD (the row loop is a chain of short scalar branches and
  // needs the other waves to fill its issue slots).
  __shared__ float red[GRAD ? Q : 1][256];
  if (GRAD) {
#pragma unroll
    for (int q = 1; q < Q; ++q) red[q][threadIdx.x] = 0.f;
  }
  double tot0 = 0.0;  // running total of the per-tile float sums
  float rows = 0.f;   // rows this lane has added to tot0

  const int64_t ntiles = (n + 63) / 64, tstep = (int64_t)gridDim.x * 4;
  int64_t t = (int64_t)blockIdx.x * 4 + wave;
  float ynext = (t < ntiles && t * 64 + lane < n) ? y[t * 64 + lane] : 0.f;
  for (; t < ntiles; t += tstep) {
    const float yv = ynext;
    const int64_t inext = (t + tstep) * 64 + lane;
    ynext = inext < n ? y[inext] : 0.f;   // rows of the next tile are in flight while this one is evaluated
    const unsigned maskv = nbins > 0 ? table.lookup(yv, tab) : (1u << K) - 1u;
    float acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = 0.f;
    auto row = [&](int j, float keep) {
      // rows RS*j .. RS*j+RS-1 of the tile are evaluated together, one per row subset of the wave
      unsigned act = 0;
#pragma unroll
      for (int r = 0; r < RS; ++r) act |= (unsigned)__builtin_amdgcn_readlane((int)maskv, RS * j + r);
      const float yj = CP == 64 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yv), j))
                                : __shfl(yv, rsub + RS * j, 64);
      mix_row<K, GRAD, true>(yj, keep, act, p, acc);
    };
    if (t * 64 + 64 <= n) {
      BVI_PRAGMA(unroll BVI_GMM_UNROLL)
      for (int j = 0; j < CP; ++j) row(j, 1.f);
      rows += (float)CP;
    } else {
      for (int j = 0; j < CP; ++j) {
        const float keep = t * 64 + rsub + RS * j < n ? 1.f : 0.f;
        row(j, keep);
        rows += keep;
      }
    }
    tot0 += (double)acc[0];
    if (GRAD) {
#pragma unroll
      for (int q = 1; q < Q; ++q) red[q][threadIdx.x] += acc[q];
    }
  }
  red[0][threadIdx.x] = (float)(tot0 - (double)rows * (double)kMixRowLevel);
  if (CP < 64) {   // add the row subsets of a wave into its first CP lanes (each thread rewrites its own column)
#pragma unroll
    for (int q = 0; q < (GRAD ? Q : 1); ++q) red[q][threadIdx.x] = bvi_reduce_over_rowsubs<CP>(red[q][threadIdx.x]);
  }
  __syncthreads();
  mix_store_partials<K, CP, GRAD>(red, kreal, blockIdx.y * CP, C, partials);
}

// minimum (MAX = false) or maximum over the wave, as a wave-uniform value; NaNs are ignored unless all are NaN
template <bool MAX>
__device__ __forceinline__ float mix_wave_extreme(float v) {
  auto comb = [](float a, float b) { return MAX ? fmaxf(a, b) : fminf(a, b); };
  auto dpp = [](float x, auto ctrl, auto rows) {   // lanes the pattern does not reach keep their own value
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x),
                                                                 decltype(ctrl)::value, decltype(rows)::value, 0xF, false));
  };
  using std::integral_constant;
  v = comb(v, dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xF>{}));    // lane ^ 1
  v = comb(v, dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xF>{}));    // lane ^ 2
  v = comb(v, dpp(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xF>{}));   // 7 - lane within eight
  v = comb(v, dpp(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xF>{}));   // 15 - lane within sixteen
  v = comb(v, dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xA>{}));   // rows 1, 3 <- lane 15 of the row before
  v = comb(v, dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xC>{}));   // rows 2, 3 <- lane 31
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Few chains per group (CG = 1, 2 or 4): every lane streams rows of its own with 16-byte loads and carries the
// sums of all CG chains in registers; the chains' parameters are wave-uniform (scalar registers).  The shuffle
// mapping above issues one 4-byte load per lane and iteration, which leaves too little in flight when the
// per-row work is small (one or a few chains).  Lanes hold different rows here, so a component can be left out
// only if it is out of reach (exponent below -kMixSkip, as in MixTable) of all 256 rows a wave has loaded: the
// wave takes the minimum and maximum of those rows (12 lane-crossing operations) and lane g K + k compares them
// with the reach of component k of chain g -- one ballot gives the active sets of all chains.  That happens when
// neighbouring rows hold similar values, i.e. when the caller has stored the rows in ascending order (the host
// layer does; any order is correct).  A wave whose rows span every component takes the branch-free form.
template <int K, int CG, bool GRAD>
__global__ __launch_bounds__(256) void gmm1d_rows_kernel(const float* __restrict__ y, int64_t n, int kreal,
                                                         const float* __restrict__ theta, int C,
                                                         float* __restrict__ partials) {
  constexpr int Q = 1 + 3 * K;
  const int c0 = blockIdx.y * CG;
  MixParams<K> p[CG];
#pragma unroll
  for (int g = 0; g < CG; ++g)   // padding chains: results dropped
    mix_load_params<K>(theta + (int64_t)(c0 + g < C ? c0 + g : 0) * 3 * kreal, kreal, p[g]);
  float acc[CG][Q];
#pragma unroll
  for (int g = 0; g < CG; ++g)
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[g][q] = 0.f;
  float rows = 0.f;
  auto row = [&](float yj) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < CG; ++g) mix_row<K, GRAD, false>(yj, 1.f, 0u, p[g], acc[g]);
    rows += 1.f;
  };
  // reach of this lane's (chain, component): [lo, hi] is where the exponent is above -kMixSkip.  Lanes beyond
  // CG * K hold an empty interval; non-finite parameters give NaN, which compares as "in reach".
  __shared__ float reach_lo[64], reach_hi[64];
  if (threadIdx.x < 64) { reach_lo[threadIdx.x] = INFINITY; reach_hi[threadIdx.x] = -INFINITY; }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int g = 0; g < CG; ++g)
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (k < kreal) {
          const float r = sqrtf(fmaxf(kMixSkip + p[g].Bp[k], 0.f)) / p[g].isg[k] * 1.0001f;
          reach_lo[g * K + k] = p[g].mu[k] - r;
          reach_hi[g * K + k] = p[g].mu[k] + r;
        }
  }
  __syncthreads();
  const float my_lo = reach_lo[threadIdx.x & 63], my_hi = reach_hi[threadIdx.x & 63];
  constexpr unsigned long long kAll = CG * K == 64 ? ~0ull : ((1ull << (CG * K)) - 1ull);
  unsigned long long all_real = 0;
#pragma unroll
  for (int g = 0; g < CG; ++g) all_real |= (unsigned long long)((1u << kreal) - 1u) << (g * K);
  // With the gradient and four chains the branch-free four-row form and the skipping form together need more
  // registers than there are: those launches keep the skipping form only (rows in no particular order then pay its
  // eight tests per chain and group of four rows with every bit set, ~10 %).
  constexpr bool kSkipOnly = GRAD && CG > 1;
  bool probing = true;       // wave-uniform: cleared by the first group of rows that spans every component
  auto rows4 = [&](const f32x4& v) __attribute__((always_inline)) {
    unsigned long long act = all_real;
    if (probing) {
      const float lmin = fminf(fminf(v.x, v.y), fminf(v.z, v.w)), lmax = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
      const float ymin = mix_wave_extreme<false>(lmin), ymax = mix_wave_extreme<true>(lmax);
      act = __builtin_amdgcn_ballot_w64(!(my_hi < ymin || my_lo > ymax)) & kAll;
      if ((act & all_real) == all_real) probing = false;       // rows in no particular order: stop looking
    }
    if (!kSkipOnly && (act & all_real) == all_real) {
      row(v.x); row(v.y); row(v.z); row(v.w);
      return;
    }
#pragma unroll
    for (int g = 0; g < CG; ++g) {
      const unsigned ag = (unsigned)(act >> (g * K)) & ((1u << K) - 1u);
      if (kSkipOnly) {     // two rows per component test: half the temporaries, three waves per SIMD instead of two
        const float ya[2] = {v.x, v.y}, yb[2] = {v.z, v.w};
        mix_rows4_skip<K, GRAD, 2>(ya, ag, p[g], acc[g]);
        mix_rows4_skip<K, GRAD, 2>(yb, ag, p[g], acc[g]);
      } else {
        const float yr[4] = {v.x, v.y, v.z, v.w};
        mix_rows4_skip<K, GRAD>(yr, ag, p[g], acc[g]);
      }
    }
    rows += 4.f;
  };
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  int64_t head = (int64_t)(((16 - ((uintptr_t)y & 15)) & 15) >> 2);
  if (head > n) head = n;
  const f32x4* __restrict__ y4 = reinterpret_cast<const f32x4*>(y + head);
  const int64_t n4 = (n - head) >> 2, tail0 = head + 4 * n4;
  int64_t i = tid;
  // whole waves only (the lane-crossing steps of rows4 want all 64 lanes): the wave's last lane decides
  for (; i + (63 - (threadIdx.x & 63)) + nth < n4; i += 2 * nth) {
    const f32x4 v0 = y4[i], v1 = y4[i + nth];
    rows4(v0);
    rows4(v1);
  }
  for (; i < n4; i += nth) {
    const f32x4 v = y4[i];
    row(v.x); row(v.y); row(v.z); row(v.w);
  }
  if (tid < head) row(y[tid]);
  if (tid < n - tail0) row(y[tail0 + tid]);
#pragma unroll
  for (int g = 0; g < CG; ++g) acc[g][0] = (float)((double)acc[g][0] - (double)rows * (double)kMixRowLevel);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float red[GRAD ? Q : 1][256];
#pragma unroll
  for (int g = 0; g < CG; ++g)
#pragma unroll
    for (int q = 0; q < (GRAD ? Q : 1); ++q) {
      const float v = bvi_reduce_over_rowsubs<1>(acc[g][q]);
      if (lane == 0) red[q][64 * wave + g] = v;
    }
  __syncthreads();
  mix_store_partials<K, CG, GRAD>(red, kreal, c0, C, partials);
}

// one thread (WAVE = false) or one wave (WAVE = true) per gradient entry
template <bool WAVE>
__global__ void gmm1d_finalize(const float* __restrict__ partials, int nb, int64_t n, int k,
                               const float* __restrict__ theta, int C, float* __restrict__ logp,
                               float* __restrict__ grad, const double* __restrict__ ref) {
  const int idx = WAVE ? (blockIdx.x * blockDim.x + threadIdx.x) >> 6 : blockIdx.x * blockDim.x + threadIdx.x;
  const int q3 = 3 * k, qreal = 1 + 3 * k;
  if (idx >= C * q3) return;
  const int c = idx / q3, j = idx - c * q3;  // j indexes the gradient entry (w | mu | scale)
  const int part = j / k, kk = j - part * k;
  if (!grad && j != 0) return;   // a value-only launch wrote L2 only
  const float* th = theta + (int64_t)c * q3;
  const float* pc = partials + (int64_t)c * qreal;
  const int64_t stride = (int64_t)C * qreal;
  const double a0 = grad ? bvi_sum_partials<WAVE>(pc + 1 + kk, nb, stride) : 0.0;
  double ax = 0.0, l2 = 0.0;
  if (part == 1) ax = bvi_sum_partials<WAVE>(pc + 1 + k + kk, nb, stride);
  if (part == 2) ax = bvi_sum_partials<WAVE>(pc + 1 + 2 * k + kk, nb, stride);
  if (j == 0) l2 = bvi_sum_partials<WAVE>(pc, nb, stride);
  if (WAVE && (threadIdx.x & 63) != 0) return;
  if (j == 0) {
    // The kernels sum v_i = log2 sum_k 2^(Bp_k - u'^2) with the FLOAT constants Bp_k and isg_k of mix_load_params.
    // Those define a parameter vector within float rounding of theta -- scales s~_k = sqrt(log2(e)/2) / isg_k,
    // weights proportional to 2^Bp_k s~_k -- whose log-density is exactly sum v_i - n (log2 W~ + log2(2 pi)/2),
    // W~ = sum_k 2^Bp_k s~_k in double: no constant that enters every row is rounded, so the error of the total is
    // that of the rows' own (unbiased) roundings and not n times a float epsilon.
    MixParams<16> p;
    mix_load_params<16>(th, k, p);
    double wt = 0.0;
    for (int i = 0; i < k; ++i) wt += exp2((double)p.Bp[i]) * (0.84932180028801904 / (double)p.isg[i]);
    const double nn = (double)n;
    l2 += nn * ((double)kMixRowLevel - log2(wt) - 0.5 * 2.6514961294723187);   // log2(2 pi)
    logp[c] = (float)(l2 * 0.6931471805599453 - (ref ? *ref : 0.0));
  }
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

// Row blocks of a few-chain launch: as many as are resident at once (the kernel's own occupancy x the CUs, shared out
// over the chain groups), so that the grid is one full round of workgroups of equal work.  With the fixed 1024 row
// blocks a kernel that fits 3 workgroups per CU ran a full round and a third of one (N = 100M, K = 8, one chain with
// gradient: 0.250 -> 0.223 ms; four chains: 0.656 -> 0.619).  Never more than `cap` (what the workspace was sized for).
template <typename Kern>
static int gmm1d_resident_blocks(Kern kern, int groups, int cap) {
  static int per_cu = 0, cus = 0;            // one pair per instantiation of this template, i.e. per kernel
  if (per_cu == 0) {
    int dev = 0, nblk = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus < 1)
      cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, kern, 256, 0) != hipSuccess || nblk < 1) nblk = 1;
    per_cu = nblk;
  }
  const int nb = per_cu * cus / (groups < 1 ? 1 : groups);
  return nb < 1 ? 1 : (nb > cap ? cap : nb);
}

// Launches the streaming kernel for `c` chains in groups of `cp`; returns the number of row blocks it used.
template <int K, bool GRAD>
static int gmm1d_dispatch_cp(int cp, int nb, hipStream_t s, const float* y, int64_t n, int k, const float* theta, int c,
                             float* ws) {
  dim3 grid(nb, bvi_ceil_div(c, cp));
  // bins of the component table: building it costs a block ~8 instructions per bin and component, so it is sized
  // to the rows a block streams (none at all for short inputs: every component is then evaluated)
  const int64_t rows_per_block = n / grid.x;
  const int nbins = rows_per_block >= 16384 ? kMixMaxBins : rows_per_block >= 2048 ? 128 : 0;
  constexpr int KR = K <= 8 ? K : 8;         // the few-chain kernels exist for K <= 8
  if (cp == 1) {
    auto kern = gmm1d_rows_kernel<K, 1, GRAD>;
    if (n >= (int64_t)nb * 1024) grid.x = gmm1d_resident_blocks(kern, grid.y, nb);
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, y, n, k, theta, c, ws);
  } else if (cp == 4 && K <= 8 && GRAD) {
    // with the gradient, two chains per lane group (four need more registers than a wave has: one wave per SIMD)
    auto kern = gmm1d_rows_kernel<KR, 2, GRAD>;
    grid.y = bvi_ceil_div(c, 2);
    if (n >= (int64_t)nb * 1024) grid.x = gmm1d_resident_blocks(kern, grid.y, nb);
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, y, n, k, theta, c, ws);
  } else if (cp == 4 && K <= 8) {
    auto kern = gmm1d_rows_kernel<KR, 4, GRAD>;
    if (n >= (int64_t)nb * 1024) grid.x = gmm1d_resident_blocks(kern, grid.y, nb);
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, y, n, k, theta, c, ws);
  } else if (cp == 4) {
    hipLaunchKernelGGL((gmm1d_kernel<K, 4, GRAD>), grid, dim3(256), 0, s, y, n, k, theta, c, nbins, ws);
  } else if (cp == 16) {
    hipLaunchKernelGGL((gmm1d_kernel<K, 16, GRAD>), grid, dim3(256), 0, s, y, n, k, theta, c, nbins, ws);
  } else {
    hipLaunchKernelGGL((gmm1d_kernel<K, 64, GRAD>), grid, dim3(256), 0, s, y, n, k, theta, c, nbins, ws);
  }
  return (int)grid.x;
}

template <bool GRAD>
static int gmm1d_dispatch_k(int cp, int nb, hipStream_t s, const float* y, int64_t n, int k, const float* theta, int c,
                            float* ws) {
  if (k <= 2) return gmm1d_dispatch_cp<2, GRAD>(cp, nb, s, y, n, k, theta, c, ws);
  if (k <= 4) return gmm1d_dispatch_cp<4, GRAD>(cp, nb, s, y, n, k, theta, c, ws);
  if (k <= 8) return gmm1d_dispatch_cp<8, GRAD>(cp, nb, s, y, n, k, theta, c, ws);
  return gmm1d_dispatch_cp<16, GRAD>(cp, nb, s, y, n, k, theta, c, ws);
}

int bvi_launch_gmm1d(const float* y, int64_t n, int k, const float* theta, int c, float* logp, float* grad,
                     float* ws, hipStream_t s, const double* logp_ref) {
  if (k < 1 || k > 16) return BVI_ERR_UNSUPPORTED;
  const int cp = bvi_chain_pack(c);
  const int nb = grad ? gmm1d_dispatch_k<true>(cp, gmm1d_blocks(n), s, y, n, k, theta, c, ws)
                      : gmm1d_dispatch_k<false>(cp, gmm1d_blocks(n), s, y, n, k, theta, c, ws);
  BVI_HIP_CHECK_LAUNCH();
  const int tot = c * 3 * k;
  if (nb > 32)
    hipLaunchKernelGGL(gmm1d_finalize<true>, dim3(bvi_ceil_div(tot, 4)), dim3(256), 0, s, (const float*)ws, nb, n, k,
                       theta, c, logp, grad, logp_ref);
  else
    hipLaunchKernelGGL(gmm1d_finalize<false>, dim3(bvi_ceil_div(tot, 128)), dim3(128), 0, s, (const float*)ws, nb, n,
                       k, theta, c, logp, grad, logp_ref);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}
