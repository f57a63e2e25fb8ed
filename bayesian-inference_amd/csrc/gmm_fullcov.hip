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
#include <stdlib.h>
#include <type_traits>
#include "bvi_common.h"

namespace {

constexpr int DP = 32;        // padded dimension
constexpr int LS = 34;        // LDS row stride in floats: conflict-free ds_read_b32 operand reads
constexpr int TR = 16;        // rows per tile

__global__ void fullcov_prepare(const float* __restrict__ theta, int C, int K, int D,
                                const float* __restrict__ center, float* __restrict__ Ainv,
                                float* __restrict__ bvec, float* __restrict__ blo, float* __restrict__ cst) {
  // blo: optional low part of b (b = bvec + blo to ~2^-48): b is common to all rows of a component, so its
  // rounding error would add up coherently over the rows
  // center: optional per-column shift the streaming kernel applies to the data (b = A (mu - center))
  // one block of DP threads per (chain, component)
  const int c = blockIdx.x / K, k = blockIdx.x % K, j = threadIdx.x;
  const int P = K + K * D + K * D * D;
  const float* th = theta + (int64_t)c * P;
  const float* mu = th + K + k * D;
  const float* L = th + K + K * D + k * D * D;
  __shared__ double A[DP][DP + 1];
  __shared__ __attribute__((aligned(16))) double Ls[DP][DP];      // L, staged once; row i is read by every lane alike
  __shared__ double invd[DP];
#pragma unroll 8
  for (int i = 0; i < DP; ++i) Ls[i][j] = (i < D && j < D) ? (double)L[i * D + j] : (i == j ? 1.0 : 0.0);
  invd[j] = j < D ? 1.0 / (double)L[j * D + j] : 1.0;
  __syncthreads();
  {
    // column j of L^-1 by forward substitution, the column in registers and every loop unrolled: one chain of
    // D^2 / 2 dependent multiply-adds per thread, with nothing else on it (as a loop over LDS it carried two LDS
    // round trips and a division per step: 39 us per call at D = 32).  Entries above the diagonal come out as the
    // zeros they are, so all threads run the same code; the padding of L is the identity.
    double a[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) {
      double s = (i == j) ? 1.0 : 0.0;
#pragma unroll
      for (int m = 0; m < i; m += 2) {
        const double2 l2 = *reinterpret_cast<const double2*>(&Ls[i][m]);
        s = fma(-l2.x, a[m], s);
        if (m + 1 < i) s = fma(-l2.y, a[m + 1], s);
      }
      a[i] = s * invd[i];
      A[i][j] = (i < D && j < D) ? a[i] : 0.0;
    }
  }
  __syncthreads();
  float* Ao = Ainv + ((int64_t)c * K + k) * DP * DP;
#pragma unroll 8
  for (int i = 0; i < DP; ++i) Ao[i * DP + j] = (float)A[i][j];
  // b = A (mu - center) with the A that the streaming kernel uses (rounded to f32): then u = A_f32 (y - mu) exactly,
  // and the rounding of A multiplies the zero-mean (y - mu) instead of the common offset (mu - center)
  __shared__ double dmu[DP];
  dmu[j] = j < D ? (double)mu[j] - (center ? (double)center[j] : 0.0) : 0.0;
  __syncthreads();
  double b = 0.0;
#pragma unroll 8
  for (int m = 0; m < DP; ++m) b += (double)(float)A[j][m] * dmu[m];
  bvec[((int64_t)c * K + k) * DP + j] = (float)b;
  if (blo) blo[((int64_t)c * K + k) * DP + j] = (float)(b - (double)(float)b);
  // constant of the component: every lane takes one logarithm and one weight, summed over the 32 lanes in a fixed order
  double ld = j < D ? log(Ls[j][j]) : 0.0, wsum = j < K ? (double)th[j] : 0.0;
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) {
    ld += __shfl_xor(ld, off, 64);
    wsum += __shfl_xor(wsum, off, 64);
  }
  if (j == 0) cst[c * K + k] = (float)(log((double)th[k]) - log(wsum) - ld - 0.5 * D * BVI_LN2PI_D);
}

// For the far-component test (fullcov_bounds), per chain and component:
//   anorm  = an upper bound of the spectral norm of A: |A|_2^2 = lambda_max(M), M = A^T A, and lambda_max(M)^(2^p) =
//            lambda_max(M^(2^p)) <= |M^(2^p)|_F, which is within DP^(1/2) of it -- p = kNormSquarings squarings in
//            float, each normalised by its largest entry: the bound on |A|_2 is within DP^(1/2^(p+2)) = 1.4 % of the
//            norm (three squarings left 11 %, which goes straight into the radius term of every tile's bounds);
//   lnorm  = the same for L = A^-1 (so that |A v| >= |v| / lnorm);
//   mudist = |mu_k - center|.
// One workgroup of 256 threads per (chain, component): thread (rg, j) owns rows rg, rg + 8, ... of column j; the two
// matrices go through the same steps side by side.
constexpr int kNormSquarings = 6;
__global__ __launch_bounds__(256) void fullcov_anorm(const float* __restrict__ theta, int K, int D,
                                                     const float* __restrict__ center, const float* __restrict__ Ainv,
                                                     float* __restrict__ anorm, float* __restrict__ mudist,
                                                     float* __restrict__ lnorm) {
  const int c = blockIdx.x / K, k = blockIdx.x % K, tid = threadIdx.x, j = tid & (DP - 1), rg = tid / DP;
  const int P = K + K * D + K * D * D;
  const float* mu = theta + (int64_t)c * P + K + k * D;
  __shared__ float Af[2][DP][DP + 1], Mq[2][2][DP][DP + 1];       // [A | L], [A | L][buffer]
  __shared__ float red[2][4];
  // sums or maxima of two values over the workgroup (NaN-blind maxima on purpose: the final sums carry a NaN through)
  auto block_reduce2 = [&](float& v0, float& v1, bool take_max) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float o0 = __shfl_xor(v0, off, 64), o1 = __shfl_xor(v1, off, 64);
      v0 = take_max ? fmaxf(v0, o0) : v0 + o0;
      v1 = take_max ? fmaxf(v1, o1) : v1 + o1;
    }
    __syncthreads();
    if ((tid & 63) == 0) { red[0][tid >> 6] = v0; red[1][tid >> 6] = v1; }
    __syncthreads();
    v0 = take_max ? fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]))
                  : (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    v1 = take_max ? fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]))
                  : (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  };
  const float* Ak = Ainv + ((int64_t)c * K + k) * DP * DP;
  const float* L = theta + (int64_t)c * P + K + K * D + (int64_t)k * D * D;
  for (int i = rg; i < DP; i += 8) {
    Af[0][i][j] = Ak[i * DP + j];
    Af[1][i][j] = (i < D && j <= i) ? L[i * D + j] : 0.f;          // the lower triangle, as the model reads it
  }
  float md2 = (rg == 0 && j < D) ? (mu[j] - center[j]) * (mu[j] - center[j]) : 0.f, unused = 0.f;
  block_reduce2(md2, unused, false);                                // (its barriers also publish Af)
  if (tid == 0) mudist[c * K + k] = sqrtf(md2);
  // Every product below: thread (rg, j) takes entries (rg + 8 r, j), r < 4, of both matrices -- one read of column j
  // per step m for the four rows, the rows' own entries as broadcast reads -- keeps them in registers over the
  // reduction that finds the largest entry, and stores them divided by it (the next product then needs no scaling).
  float v[2][4], cmax[2] = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[q][r] = 0.f;
#pragma unroll 8
    for (int m = 0; m < DP; ++m) {
      const float bcol = Af[q][m][j];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[q][r] = fmaf(Af[q][m][rg + 8 * r], bcol, v[q][r]);      // M = A^T A
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) cmax[q] = fmaxf(cmax[q], fabsf(v[q][r]));
  }
  float scale[2][kNormSquarings + 1];
  int cur = 0;
#pragma unroll
  for (int sq = 0; sq <= kNormSquarings; ++sq) {
    block_reduce2(cmax[0], cmax[1], true);
    scale[0][sq] = cmax[0];
    scale[1][sq] = cmax[1];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float inv = scale[q][sq] > 0.f ? 1.f / scale[q][sq] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) Mq[q][cur][rg + 8 * r][j] = v[q][r] * inv;
    }
    __syncthreads();
    if (sq == kNormSquarings) break;
    cmax[0] = cmax[1] = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[q][r] = 0.f;
#pragma unroll 8
      for (int m = 0; m < DP; ++m) {
        const float bcol = Mq[q][cur][m][j];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[q][r] = fmaf(Mq[q][cur][rg + 8 * r][m], bcol, v[q][r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) cmax[q] = fmaxf(cmax[q], fabsf(v[q][r]));
    }
    cur ^= 1;
  }
  float fro[2] = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) fro[q] = fmaf(Mq[q][cur][rg + 8 * r][j], Mq[q][cur][rg + 8 * r][j], fro[q]);
  block_reduce2(fro[0], fro[1], false);
  if (tid == 0) {
    // lambda(M) <= s0 sqrt(s1 sqrt(s2 ... sqrt(s_p |N_p|_F)))
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float lam = scale[q][kNormSquarings] * sqrtf(fro[q]);
#pragma unroll
      for (int sq = kNormSquarings - 1; sq >= 0; --sq) lam = scale[q][sq] * sqrtf(lam);
      (q == 0 ? anorm : lnorm)[c * K + k] = sqrtf(lam) * 1.0001f;
    }
  }
}

// One pass over Y for both kinds of statistics.  Per 64-row tile: centre (column means) and radius (largest distance
// of a row from the centre), raw coordinates: tstat[tile][0 .. DP-1] = centre, [DP] = radius (tstat == nullptr: not
// wanted).  Per column: sum (double), min, max over the rows of this workgroup's tiles, psum / pmin / pmax[block][DP]
// (the first stage of the column statistics; fullcov_colstats_final is the second).  One wave per tile; lane
// (ln, g) = (lane & 15, lane >> 4) holds columns 8 g .. 8 g + 7 of rows ln, ln + 16, ln + 32, ln + 48 (a row is one
// contiguous 128 B across g); column sums run down the 16 lanes of a DPP row, a row's distance across g.  A
// workgroup's four waves stride over the tiles.
__device__ __forceinline__ float tilestats_row16_sum(float v) {       // sum over the 16 lanes of a DPP row, in every lane
#define BVI_TS_STEP(ctrl) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
  BVI_TS_STEP(0xB1);       // quad_perm [1, 0, 3, 2]
  BVI_TS_STEP(0x4E);       // quad_perm [2, 3, 0, 1]
  BVI_TS_STEP(0x124);      // row_ror 4
  BVI_TS_STEP(0x128);      // row_ror 8
#undef BVI_TS_STEP
  return v;
}
__global__ __launch_bounds__(256) void fullcov_tilestats(const float* __restrict__ Y, int64_t n, int D, int64_t ntiles,
                                                         float* __restrict__ tstat, double* __restrict__ psum,
                                                         float* __restrict__ pmin, float* __restrict__ pmax) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ln = lane & 15, g = lane >> 4;
  const bool vec_ok = D == DP && (((uintptr_t)Y & 15) == 0);
  double csum[8];                                      // lanes with ln == 0: columns 8 g + e
  float mn[8], mx[8];                                  // this lane's rows
#pragma unroll
  for (int e = 0; e < 8; ++e) { csum[e] = 0.0; mn[e] = INFINITY; mx[e] = -INFINITY; }
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t left = n - tile * 64;
    const float icnt = 1.f / (float)(left < 64 ? left : 64);
    float y[4][8];
    bool valid[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int64_t row = tile * 64 + 16 * rt + ln;
      valid[rt] = row < n;
      if (vec_ok) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (valid[rt]) {
          a = *reinterpret_cast<const f32x4*>(Y + row * DP + 8 * g);
          b = *reinterpret_cast<const f32x4*>(Y + row * DP + 8 * g + 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { y[rt][e] = a[e]; y[rt][4 + e] = b[e]; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) y[rt][e] = (valid[rt] && 8 * g + e < D) ? Y[row * D + 8 * g + e] : 0.f;
      }
    }
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        if (valid[rt]) { mn[e] = fminf(mn[e], y[rt][e]); mx[e] = fmaxf(mx[e], y[rt][e]); }
      const float sm = tilestats_row16_sum((y[0][e] + y[1][e]) + (y[2][e] + y[3][e]));
      m[e] = sm * icnt;
      csum[e] += (double)sm;                           // (kept by the lanes with ln == 0)
    }
    if (tstat && ln == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) tstat[tile * (DP + 1) + 8 * g + e] = m[e];
    }
    float dmax = 0.f;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      float d2 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dv = valid[rt] ? y[rt][e] - m[e] : 0.f;
        d2 = fmaf(dv, dv, d2);
      }
      d2 += __shfl_xor(d2, 16, 64);
      d2 += __shfl_xor(d2, 32, 64);
      dmax = (d2 > dmax || d2 != d2) ? d2 : dmax;        // NaN wins: such a tile keeps every component
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
      const float o = __shfl_xor(dmax, off, 64);
      dmax = (o > dmax || o != o) ? o : dmax;
    }
    if (lane == 0 && tstat) tstat[tile * (DP + 1) + DP] = sqrtf(dmax);
  }
  // column minima / maxima over the wave's rows (down the 16 lanes of a DPP row), then over the four waves
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
      mn[e] = fminf(mn[e], __shfl_xor(mn[e], off, 64));
      mx[e] = fmaxf(mx[e], __shfl_xor(mx[e], off, 64));
    }
  __shared__ double ssum[4][DP];
  __shared__ float smn[4][DP], smx[4][DP];
  if (ln == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[wave][8 * g + e] = csum[e]; smn[wave][8 * g + e] = mn[e]; smx[wave][8 * g + e] = mx[e]; }
  }
  __syncthreads();
  if (threadIdx.x < DP) {
    const int d = threadIdx.x;
    double t = ssum[0][d];
    float a = smn[0][d], b = smx[0][d];
    for (int w = 1; w < 4; ++w) { t += ssum[w][d]; a = fminf(a, smn[w][d]); b = fmaxf(b, smx[w][d]); }
    psum[blockIdx.x * DP + d] = d < D ? t : 0.0;
    pmin[blockIdx.x * DP + d] = a;
    pmax[blockIdx.x * DP + d] = b;
  }
}

// Spheres around groups of 64 consecutive tiles: centre = mean of the tiles' centres, radius = the largest
// (distance of a tile's centre + that tile's radius).  Same layout as tstat.  One wave per group, lane = tile.
__global__ __launch_bounds__(64) void fullcov_groupstats(const float* __restrict__ tstat, int64_t ntiles,
                                                         float* __restrict__ gstat) {
  const int lane = threadIdx.x;
  const int64_t t = (int64_t)blockIdx.x * 64 + lane;
  const bool tv = t < ntiles;
  const float* ts = tstat + (tv ? t : 0) * (DP + 1);
  const float nv = (float)__builtin_popcountll(__builtin_amdgcn_ballot_w64(tv));
  float e2 = 0.f;
  float* out = gstat + (int64_t)blockIdx.x * (DP + 1);
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    const float x = ts[d];
    float v = tv ? x : 0.f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    const float m = v / nv;
    if (lane == d) out[d] = m;
    const float e = x - m;
    e2 = fmaf(e, e, e2);
  }
  float rs = tv ? sqrtf(e2) + ts[DP] : 0.f;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float o = __shfl_xor(rs, off, 64);
    rs = (o > rs || o != o) ? o : rs;                    // NaN wins: such a group keeps every component
  }
  if (lane == 0) out[DP] = rs;
}

// The bounds of fullcov_bounds for the groups of 64 tiles, one (chain, component) per workgroup and one group per
// lane (the groups are few: looping over the components per group would leave the chip empty):
// gb[chain][group][k] = (upper, lower) bound of component k's log-density over the group's rows; anything
// non-finite becomes (+inf, -inf), which keeps the component and never serves as the threshold.
__global__ __launch_bounds__(64) void fullcov_group_bounds(const float* __restrict__ gstat, int64_t ngroups, int K,
                                                           const float* __restrict__ Ainv, const float* __restrict__ bvec,
                                                           const float* __restrict__ cst, const float* __restrict__ anorm,
                                                           const float* __restrict__ center, float2* __restrict__ gb) {
  const int c = blockIdx.y, k = blockIdx.z, lane = threadIdx.x;
  const int64_t t = (int64_t)blockIdx.x * 64 + lane;
  __shared__ __attribute__((aligned(16))) float As[DP * DP];
  __shared__ float bs[DP];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<f32x4*>(&As[4 * (lane + 64 * q)]) =
        *reinterpret_cast<const f32x4*>(Ainv + ((int64_t)c * K + k) * DP * DP + 4 * (lane + 64 * q));
  if (lane < DP) bs[lane] = bvec[((int64_t)c * K + k) * DP + lane];
  __syncthreads();
  const bool tv = t < ngroups;
  const float* ts = gstat + (tv ? t : 0) * (DP + 1);
  float dm[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) dm[d] = ts[d] - center[d];
  const float r = ts[DP];
  float d2 = 0.f;
#pragma unroll
  for (int i = 0; i < DP; ++i) {
    float sacc = -bs[i];
#pragma unroll
    for (int j4 = 0; j4 <= i; j4 += 4) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(&As[i * DP + j4]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (j4 + e <= i) sacc = fmaf(a4[e], dm[j4 + e], sacc);
    }
    d2 = fmaf(sacc, sacc, d2);
  }
  const float dk = sqrtf(d2), slack = anorm[c * K + k] * r * 1.001f + 1e-3f;
  const float lo = fmaxf(dk - slack, 0.f), hi = dk + slack;
  const float ck = cst[c * K + k];
  float ub = ck - 0.5f * lo * lo, lb = ck - 0.5f * hi * hi;
  if (!(fabsf(ub) < INFINITY) || !(fabsf(lb) < INFINITY)) {
    // a component of weight 0 (constant -inf) is out of reach everywhere; anything else that is not finite stays in
    ub = ck > -INFINITY || ck != ck ? INFINITY : -INFINITY;
    lb = -INFINITY;
  }
  if (tv) gb[((int64_t)c * ngroups + t) * K + k] = make_float2(ub, lb);
}

// Which components a sphere of rows (a tile, or a group of 64 tiles) can leave out.  For a row y in the sphere and
// u_k(y) = A_k (y - mu_k):
//   | |u_k(y)| - |u_k(m)| | <= |A_k|_2 |y - m| <= anorm_k r      (m, r: the sphere's centre and radius)
// so a_k(y) = cst_k - |u_k(y)|^2 / 2 lies between cst_k - hi_k^2 / 2 and cst_k - lo_k^2 / 2 with
// hi / lo = |u_k(m)| +- anorm_k r.  A component whose upper bound is more than 32 below the largest lower bound has
// responsibilities below e^-32 = 2^-46 for every row of the sphere: bit k of mask[chain][sphere] is cleared and the
// streaming kernels do not evaluate it there (its terms are below the 2^-40 the raw-sums phase keeps anyway).
// Anything non-finite keeps the bit set.  One thread per tile, one wave per group of 64 tiles, in three steps of
// rising cost, each only for what the one before left open:
//   1. without any product: |u_k(m)| <= anorm_k (|m - centre| + |mu_k - centre|) bounds lo_k from above and
//      hi_j >= anorm_j r bounds the largest lower bound from above; if that already keeps every component (rows in
//      no particular order: r is the size of the whole dataset) the tile keeps what its group kept (`parent`: the
//      group's bounds from fullcov_group_bounds; with rows stored cluster by cluster most groups keep one component,
//      which settles all their tiles);
//   2. from the distance to the component's mean alone (D multiplications per component):
//      |m - mu_k| / lnorm_k <= |u_k(m)| <= anorm_k |m - mu_k|.  A tile in the middle of its cluster is left with that
//      cluster's component; a tile that straddles two clusters (one per boundary) keeps many, and stays as it is;
//   3. |u_k(m)| itself (D^2 / 2 multiplications, A_k through LDS, every lane reading the same element), only when
//      some tile of the wave is left with two to six components by step 2: neighbouring or overlapping clusters.
constexpr int kRefineMax = 6;      // step 3 runs for tiles step 2 leaves with 2 .. kRefineMax components
__global__ __launch_bounds__(64) void fullcov_bounds(const float* __restrict__ tstat, int64_t ntiles, int K, int D,
                                                      const float* __restrict__ theta,
                                                      const float* __restrict__ Ainv, const float* __restrict__ bvec,
                                                      const float* __restrict__ cst, const float* __restrict__ anorm,
                                                      const float* __restrict__ mudist, const float* __restrict__ lnorm,
                                                      const float* __restrict__ center, int C,
                                                      const float2* __restrict__ parent, int64_t nparent,
                                                      unsigned short* __restrict__ mask_out) {
  const int c = blockIdx.y, lane = threadIdx.x;
  const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
  __shared__ float aub[16][64];
  __shared__ __attribute__((aligned(16))) float As[DP * DP];
  __shared__ float bs[DP];
  __shared__ __attribute__((aligned(16))) float mus[16][DP];     // mu_k - centre
  const bool tv = t < ntiles;
  const float* ts = tstat + (tv ? t : 0) * (DP + 1);
  float dm[DP];
  float dn2 = 0.f;
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    dm[d] = ts[d] - center[d];
    dn2 = fmaf(dm[d], dm[d], dn2);
  }
  const float r = ts[DP];
  const unsigned full = (1u << K) - 1u;
  // the chain's per-component numbers and the parent's bounds: lane k holds component k's (one coalesced load each;
  // read back by readlane -- as scalar loads inside the loops they were a chain of K waits)
  const int kl = lane < K ? lane : 0;
  const float an_l = anorm[c * K + kl], ck_l = cst[c * K + kl], md_l = mudist[c * K + kl], ln_l = lnorm[c * K + kl];
  float2 pb_l = make_float2(INFINITY, -INFINITY);
  if (parent) pb_l = parent[((int64_t)c * nparent + (t - lane) / 64) * K + kl];
  // 1.
  bool keeps_all = true;
  {
    const float dn = sqrtf(dn2);
    float top = -INFINITY, low = INFINITY;            // upper bound of the largest lower bound; smallest a_ub bound
    for (int k = 0; k < K; ++k) {
      const float an = __shfl(an_l, k, 64), ck = __shfl(ck_l, k, 64);
      const float sl = an * r;
      const float lomax = fmaxf(an * (dn + __shfl(md_l, k, 64)) * 1.001f + 1e-3f - sl, 0.f);
      top = fmaxf(top, ck - 0.5f * sl * sl);
      low = fminf(low, ck - 0.5f * lomax * lomax);
    }
    keeps_all = !(low < top - 30.f);                  // (NaN: keeps everything)
  }
  unsigned sup = full;
  if (parent) {                                         // the group's mask from its components' bounds
    float gbest = lane < K ? pb_l.y : -INFINITY;
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) gbest = fmaxf(gbest, __shfl_xor(gbest, off, 64));
    gbest = __shfl(gbest, 0, 64);
    const float gth = gbest - 32.f - 1e-4f * fabsf(gbest);
    sup = (unsigned)__builtin_amdgcn_ballot_w64(lane < K && !(pb_l.x < gth));
  }
  unsigned mask = sup;
  // 2. (the same decision in every lane: the barriers below are met by all)
  if (__builtin_popcount(sup) > 1 && __builtin_amdgcn_ballot_w64(tv && !keeps_all) != 0) {
    const int P = K + K * D + K * D * D;
    for (int e = lane; e < K * DP; e += 64) {
      const int k = e / DP, d = e % DP;
      mus[k][d] = d < D ? theta[(int64_t)c * P + K + k * D + d] - center[d] : 0.f;
    }
    __syncthreads();
    float best = -INFINITY;
    unsigned open = 0;                                  // components whose cheap bounds are finite
    float ubc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      ubc[k] = INFINITY;
      if (k < K && ((sup >> k) & 1u)) {                 // (the same for the whole wave)
        float d2 = 0.f;
#pragma unroll
        for (int q = 0; q < DP / 4; ++q) {
          const f32x4 m4 = *reinterpret_cast<const f32x4*>(&mus[k][4 * q]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dv = dm[4 * q + e] - m4[e];
            d2 = fmaf(dv, dv, d2);
          }
        }
        const float dist = sqrtf(d2), an = __shfl(an_l, k, 64), ck = __shfl(ck_l, k, 64);
        const float slack = an * r * 1.001f + 1e-3f;
        const float lo = fmaxf(dist / __shfl(ln_l, k, 64) * 0.999f - slack, 0.f), hi = an * dist * 1.001f + slack;
        const float ub = ck - 0.5f * lo * lo, lb = ck - 0.5f * hi * hi;
        if ((fabsf(ub) < INFINITY) && (fabsf(lb) < INFINITY)) {
          ubc[k] = ub;
          open |= 1u << k;
          best = fmaxf(best, lb);
        } else if (!(ck > -INFINITY || ck != ck)) {
          ubc[k] = -INFINITY;                           // weight 0: out of reach everywhere
        }
      }
    }
    const float th2 = best - 32.f - 1e-4f * fabsf(best);
    unsigned m2 = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k < K && ((sup >> k) & 1u) && !(ubc[k] < th2)) m2 |= 1u << k;
    if (!keeps_all) mask = m2;
    // 3.
    const bool refine = tv && !keeps_all && __builtin_popcount(m2) >= 2 && __builtin_popcount(m2) <= kRefineMax &&
                        (m2 & ~open) == 0;
    unsigned supx = 0;
    for (int k = 0; k < K; ++k)
      if (__builtin_amdgcn_ballot_w64(refine && ((m2 >> k) & 1u)) != 0) supx |= 1u << k;
    float bestx = -INFINITY;
    bool odd = false;                      // something is not finite: keep what step 2 kept
    for (int k = 0; k < K; ++k) {
      if (!((supx >> k) & 1u)) continue;   // (the same for the whole wave)
      __syncthreads();                     // the previous component's readers are done
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(&As[4 * (lane + 64 * q)]) =
            *reinterpret_cast<const f32x4*>(Ainv + ((int64_t)c * K + k) * DP * DP + 4 * (lane + 64 * q));
      if (lane < DP) bs[lane] = bvec[((int64_t)c * K + k) * DP + lane];
      __syncthreads();
      float d2 = 0.f;
#pragma unroll
      for (int i = 0; i < DP; ++i) {
        float sacc = -bs[i];
#pragma unroll
        for (int j4 = 0; j4 <= i; j4 += 4) {
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(&As[i * DP + j4]);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j4 + e <= i) sacc = fmaf(a4[e], dm[j4 + e], sacc);
        }
        d2 = fmaf(sacc, sacc, d2);
      }
      const float dk = sqrtf(d2), slack = __shfl(an_l, k, 64) * r * 1.001f + 1e-3f;
      const float lo = fmaxf(dk - slack, 0.f), hi = dk + slack;
      const float ck = __shfl(ck_l, k, 64);
      const float ub = ck - 0.5f * lo * lo, lb = ck - 0.5f * hi * hi;
      aub[k][threadIdx.x] = ub;
      if (lb > bestx) bestx = lb;
      if (!(fabsf(ub) < INFINITY) || !(fabsf(lb) < INFINITY)) odd = true;
    }
    if (refine && !odd) {
      const float th3 = bestx - 32.f - 1e-4f * fabsf(bestx);
      unsigned m3 = 0;
      for (int k = 0; k < K; ++k)
        if (((m2 >> k) & 1u) && !(aub[k][threadIdx.x] < th3)) m3 |= 1u << k;
      mask = m3;
    }
  }
  if (tv) mask_out[(int64_t)c * ntiles + t] = (unsigned short)mask;
}

// The level a typical row's log-density sits at, from the components' constants as the kernels hold them (floats):
// the tallest component's constant minus D/2 (|u|^2 ~ D for a row of its own component).  The f16 kernel sums
// log-densities relative to it, so that the float partial sums stay random-walk sized (|log p| ~ 4.5e7 at a
// million rows of D = 32, where floats are 4 apart); fullcov_finalize adds n times the same float back in double.
__device__ __forceinline__ float fullcov_row_level(const float* __restrict__ cst, int K, int D) {
  float m = -INFINITY;
  for (int k = 0; k < K; ++k) m = fmaxf(m, cst[k]);
  return isfinite(m) ? m - 0.5f * (float)D : 0.f;
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

// ---------------------------------------------------------------------------------------------
// The same sums on the f16 matrix cores (v_mfma_f32_16x16x32_f16), operands split into f16 pieces
// as in linreg_f16.hip: the data in 2 pieces (error <= 1 f32 ulp, independent from row to row), the
// whitening matrices A_k in 3 (exact: an error in a parameter is common to all rows and would grow
// with n), products yh.a1, yl.a1, yh.a2, yh.a3.  Range normalisation by exact powers of two:
//   y'_id = (y_id - m_d) s_d   (m_d: column mean, s_d = 2^(13 - floor(log2 max_i |y_id - m_d|)); both from a
//                               statistics pass over Y per call; centring also removes the cancellation
//                               between A y and A mu for data far from the origin)
//   A''_k = A_k diag(1/s) t_k, t_k = 2^(13 - floor(log2 max |A_k diag(1/s)|))   (per chain and component)
// so z' = A'' y' = t_k A_k (y - m) and u = z' / t_k - A_k (mu_k - m).
// S2_k = sum_i rho_ik u u^T is accumulated as W^T W with w = sqrt(rho) u = exp((a - lse)/2) u split into 2
// f16 pieces (wh.wh, wl.wh, wh.wl): the accumulator registers of GEMM1 (rows in registers, dims on
// lanes) are, lane for lane, both the A and the B operand of that product for two 16-row tiles side by
// side, so S2 needs no LDS traffic at all; S0, S1 are register sums.
//
// Workgroup = one chain x a strided set of 64-row tiles, 8 waves, wave w owns components w, w + 8.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int FT = 64;        // rows per tile of the f16 kernel
constexpr int FW = 8;         // waves per workgroup

__device__ __forceinline__ float fc_pow2_scale(float amax) {
  if (!(amax > 0.f) || !isfinite(amax)) return 1.f;
  return ldexpf(1.f, min(max(13 - ilogbf(amax), -126), 126));
}

__device__ __forceinline__ void fc_split_pair(float q0, float q1, unsigned& hi, unsigned& lo) {
  asm volatile(
      "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
      "v_fma_mixlo_f16 %1, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %1, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(hi), "=&v"(lo)
      : "v"(q0), "v"(q1));
}

// step S (0..3) of the sum over the 16 lanes of a DPP row (after all four: the result in every lane)
template <int S>
__device__ __forceinline__ float row16_sum_step(float v) {
  constexpr int ctrl = S == 0 ? 0xB1 : S == 1 ? 0x4E : S == 2 ? 0x124 : 0x128;
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true));
}

// Sums of four values over the 16 lanes of a DPP row in 8 instead of 16 lane-crossing adds: the first two steps fold
// across the row's four banks of four lanes and halve the number of values each time (bank masks pick which value a
// bank keeps), the last two run inside a bank.  After step 3 every lane of bank b holds the sum of v[b].  One call
// per step so that the caller can put other work between them.  A lane-crossing read needs two wait states after the
// write of its source, which the assembler does not insert around an asm statement: PAD = true pads them with an
// s_nop, PAD = false is for callers that have at least two other instructions between the steps.
struct Row16Sum4 {
  float d0, d1;
  template <bool PAD>
  __device__ __forceinline__ void step0(const float (&v)[4]) {   // lanes L and 15 - L: banks 0, 1 keep v0, v1; 2, 3 keep v2, v3
    if (PAD) asm volatile("s_nop 1");
    asm volatile(
        "v_add_f32_dpp %0, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %4 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %5, %5 row_mirror row_mask:0xf bank_mask:0xc"
        : "=&v"(d0), "=&v"(d1)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
  }
  template <bool PAD>
  __device__ __forceinline__ void step1() {                      // lanes L and 7 - L of a half row: bank b keeps v[b]
    float e;
    if (PAD) asm volatile("s_nop 1");
    asm volatile(
        "v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa"
        : "=&v"(e)
        : "v"(d0), "v"(d1));
    d0 = e;
  }
  template <bool PAD>
  __device__ __forceinline__ void step2() {
    float e;
    if (PAD) asm volatile("s_nop 1");
    asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(e) : "v"(d0));
    d0 = e;
  }
  template <bool PAD>
  __device__ __forceinline__ void step3() {
    float e;
    if (PAD) asm volatile("s_nop 1");
    asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(e) : "v"(d0));
    d0 = e;
  }
  template <bool PAD>
  __device__ __forceinline__ void step(int k, const float (&v)[4]) {
    if (k == 0) step0<PAD>(v);
    else if (k == 1) step1<PAD>();
    else if (k == 2) step2<PAD>();
    else step3<PAD>();
  }
};

// the value of lane L ^ 1 (CTRL 0xB1), L ^ 2 (0x4E) or 7 - L within each group of eight lanes (0x141)
template <int CTRL>
__device__ __forceinline__ float lane_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// sum over the 16 lanes of a DPP row, result in every lane
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
  return v;
}

// column statistics, stage 2 (stage 1: fullcov_tilestats; one workgroup of 1024 threads = 32 groups x DP columns, each
// group folds every 32nd partial, always in the same order):
// center[d] = mean, cscale[d] = power-of-two scale of max |y - mean| (0 for a constant / unused column)
__global__ __launch_bounds__(1024) void fullcov_colstats_final(const double* __restrict__ psum,
                                                              const float* __restrict__ pmin,
                                                              const float* __restrict__ pmax, int nblk, int64_t n,
                                                              int D, float* __restrict__ center,
                                                              float* __restrict__ cscale) {
  static_assert(DP == 32, "32 groups x 32 columns");
  constexpr int NG = 32;
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  double sum = 0.0;
  float mn = INFINITY, mx = -INFINITY;
  for (int b = grp; b < nblk; b += NG) {
    sum += psum[b * DP + col];
    mn = fminf(mn, pmin[b * DP + col]);
    mx = fmaxf(mx, pmax[b * DP + col]);
  }
  __shared__ double ssum[NG][DP];
  __shared__ float smn[NG][DP], smx[NG][DP];
  ssum[grp][col] = sum; smn[grp][col] = mn; smx[grp][col] = mx;
  __syncthreads();
  if (grp != 0) return;
  for (int q = 1; q < NG; ++q) {
    sum += ssum[q][col];
    mn = fminf(mn, smn[q][col]);
    mx = fmaxf(mx, smx[q][col]);
  }
  float m = 0.f, sc = 0.f;
  if (col < D && n > 0) {
    m = (float)(sum / (double)n);
    if (!isfinite(m)) m = 0.f;
    const float amax = fmaxf(mx - m, m - mn);
    sc = amax > 0.f ? fc_pow2_scale(amax) : 0.f;
  }
  center[col] = m;
  cscale[col] = sc;
}


// Replication of the components over the waves of fullcov_single_kernel: with K <= 4 components each is given to
// R = 8, 4 or 2 waves (every wave then has work), otherwise to one.
__host__ __device__ __forceinline__ int fullcov_replicas(int K) { return K <= 1 ? 8 : K <= 2 ? 4 : K <= 4 ? 2 : 1; }

// Sorts the tiles of every (chain, row block) into 2 FW + 1 lists: list w + FW q (w < FW, q < 2) = what wave w of
// fullcov_single_kernel does for component (w % K8) + FW q (with replicated components the replicas take turns):
// first the tiles that component alone reaches, then -- once per component -- the tiles that two or three components
// reach (kFewMax; `both` false, a value-only call: under their first component only, the one that adds their
// log-densities); list 2 FW = the tiles more components reach, or none (fullcov_f16_kernel).  An entry is (tile,
// mask of the components that reach it).  tlist[(chain, block)][cap] holds the lists back to back,
// toff[(chain, block)][kListOffs]: the lists' starts, the end, one flag -- whether the masks leave out at least an
// eighth of the (tile, component) pairs of the last list: then the instantiation of fullcov_f16_kernel that looks at
// the masks takes this (chain, block), otherwise the one without that code (which is a tenth faster on full masks) --
// and, for the first 2 FW lists, where their second part starts.  tmask == nullptr: every tile is reached by all K
// components.  One wave per (chain, block), 64 tiles per step, positions from ballots: inside each part of a list the
// order is by step, then by the component's place in the tile's mask, then by tile -- the same on every run.
constexpr int kLists = 2 * FW + 1, kListOffs = kLists + 2 + 2 * FW;
constexpr int kFewOff = kLists + 2;                  // toff[kFewOff + j]: start of the second part of list j
constexpr int kFewMax = 3;
__global__ __launch_bounds__(64) void fullcov_tilelists(const unsigned short* __restrict__ tmask, int64_t ntiles, int K,
                                                        int cap, int2* __restrict__ tlist, int* __restrict__ toff,
                                                        int* __restrict__ any_single, int both) {
  const int c = blockIdx.y, b = blockIdx.x, nb = gridDim.x, lane = threadIdx.x;
  const int R = fullcov_replicas(K), K8 = FW / R;
  const unsigned kmask = (1u << K) - 1u;
  int2* seg = tlist + ((int64_t)c * nb + b) * cap;
  int* so = toff + ((int64_t)c * nb + b) * kListOffs;
  const int cnt_tiles = b < ntiles ? (int)((ntiles - b + nb - 1) / nb) : 0;
  int saved = 0;                                      // (tile, component) pairs of the last list the masks leave out
  auto mask_of = [&](int i) -> unsigned {             // i-th tile of this block (0 past the end: no list)
    if (i >= cnt_tiles) return 0u;
    return tmask ? (tmask[(int64_t)c * ntiles + b + (int64_t)i * nb] & kmask) : kmask;
  };
  auto list_of = [&](int i, int k) { return (k % FW) % K8 + K8 * (i % R) + (k >= FW ? FW : 0); };
  // first part: the list of a tile one component reaches, 2 FW for a tile of the last list, -1: neither
  auto classify = [&](int i, unsigned m) -> int {
    if (i >= cnt_tiles) return -1;
    const int pc = __builtin_popcount(m);
    if (pc == 0 || pc > kFewMax) return 2 * FW;
    return pc == 1 ? list_of(i, __builtin_ctz(m)) : -1;
  };
  // second part: the list of slot s (the s-th component of its mask) of a tile two or three components reach
  auto classify_few = [&](int i, unsigned m, int s) -> int {
    if (i >= cnt_tiles) return -1;
    const int pc = __builtin_popcount(m);
    if (pc < 2 || pc > kFewMax || s >= pc || (s > 0 && !both)) return -1;
    unsigned r = m;
    for (int q = 0; q < s; ++q) r &= r - 1;
    return list_of(i, __builtin_ctz(r));
  };
  int count[kLists], countf[2 * FW];
#pragma unroll
  for (int j = 0; j < kLists; ++j) count[j] = 0;
#pragma unroll
  for (int j = 0; j < 2 * FW; ++j) countf[j] = 0;
  for (int i0 = 0; i0 < cnt_tiles; i0 += 64) {
    const unsigned m = mask_of(i0 + lane);
    const int pc = __builtin_popcount(m);
    if (i0 + lane < cnt_tiles && (pc == 0 || pc > kFewMax)) saved += K - pc;
    const int cl = classify(i0 + lane, m);
#pragma unroll
    for (int j = 0; j < kLists; ++j) count[j] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(cl == j));
    for (int s = 0; s < kFewMax; ++s) {
      if (__builtin_amdgcn_ballot_w64(pc > s && pc >= 2 && pc <= kFewMax) == 0) break;      // (the same in every lane)
      const int cf = classify_few(i0 + lane, m, s);
#pragma unroll
      for (int j = 0; j < 2 * FW; ++j) countf[j] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(cf == j));
    }
  }
  int start[kLists], startf[2 * FW], run = 0;
#pragma unroll
  for (int j = 0; j < kLists; ++j) {
    start[j] = run;
    run += count[j];
    if (j < 2 * FW) { startf[j] = run; run += countf[j]; }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) saved += __shfl_xor(saved, off, 64);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < kLists; ++j) so[j] = start[j];
#pragma unroll
    for (int j = 0; j < 2 * FW; ++j) so[kFewOff + j] = startf[j];
    so[kLists] = run;
    so[kLists + 1] = (tmask != nullptr && saved * 8 >= K * count[kLists - 1] && saved > 0) ? 1 : 0;
    if (start[kLists - 1] > 0) *any_single = 1;       // some tile goes to fullcov_single_kernel: it needs the planes
  }
  for (int i0 = 0; i0 < cnt_tiles; i0 += 64) {
    const unsigned m = mask_of(i0 + lane);
    const int pc = __builtin_popcount(m);
    const int2 e = make_int2((int)(b + (int64_t)(i0 + lane) * nb), (int)m);
    const int cl = classify(i0 + lane, m);
#pragma unroll
    for (int j = 0; j < kLists; ++j) {
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(cl == j);
      if (cl == j) seg[start[j] + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = e;
      start[j] += __builtin_popcountll(bal);
    }
    for (int s = 0; s < kFewMax; ++s) {
      if (__builtin_amdgcn_ballot_w64(pc > s && pc >= 2 && pc <= kFewMax) == 0) break;
      const int cf = classify_few(i0 + lane, m, s);
#pragma unroll
      for (int j = 0; j < 2 * FW; ++j) {
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(cf == j);
        if (cf == j) seg[startf[j] + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = e;
        startf[j] += __builtin_popcountll(bal);
      }
    }
  }
}

// GRAD = false (value only, what random-walk Metropolis asks for): no responsibilities, no raw sums
// PRUNE: the instantiation that looks at the tiles' component masks; each (chain, row block) is done by the one its
// flag names (fullcov_tilelists), the other returns at once
template <int KQ, bool GRAD, bool PRUNE>  // components per wave (K <= FW * KQ)
__global__ __launch_bounds__(64 * FW) __attribute__((amdgpu_waves_per_eu(2, GRAD ? 2 : 4))) void fullcov_f16_kernel(
    const float* __restrict__ Y, int64_t n, int D, int K, const float* __restrict__ Ainv,
    const float* __restrict__ bvec, const float* __restrict__ blo, const float* __restrict__ cst,
    const float* __restrict__ center, const float* __restrict__ cscale, int C, float* __restrict__ partials,
    const unsigned short* __restrict__ tmask, const int2* __restrict__ tlist, const int* __restrict__ toff, int cap) {
  // LDS: two f16 images of the tile (hi, lo), 64-B rows with the 16-B chunks XOR-swizzled by row so that the
  // A-operand ds_read_b128 (16 rows x two neighbouring chunks per lane group) is conflict-free; double buffered
  __shared__ __attribute__((aligned(16))) unsigned char ybuf[2][2][FT * 64];
  __shared__ float al[FW * KQ][FT + 4];   // a_ik of the tile (rows padded: the per-row pass reads a column of it)
  __shared__ float sq[FW * KQ][FT + 4];   // sqrt(rho_ik) of the tile (0 for rows past the end)
  __shared__ float cks[FW * KQ];        // the components' constants
  __shared__ float lpw[FT];
  // actf[q][kp][w] != 0: component kp + FW q has a responsibility worth multiplying among rows 8 w .. 8 w + 7 of the tile
  __shared__ __attribute__((aligned(8))) unsigned char actf[2][FW][FW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool lead_wave = __builtin_amdgcn_readfirstlane(wave) < FW / 2;      // wave-uniform, in a scalar register
  const int ln = lane & 15, g = lane >> 4;
  const int c = blockIdx.y;
  {
    const int* so0 = toff + ((int64_t)c * gridDim.x + blockIdx.x) * kListOffs;
    if ((so0[kLists + 1] != 0) != PRUNE) return;
    if (so0[kLists - 1] == so0[kLists]) return;      // no tile of this kind: fullcov_finalize skips the block's partials
  }
  auto swz = [](int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; };

  // A''_k as B operand of GEMM1 (k = input dim, n = output dim): lane holds output dim 16 nt + ln, input dims
  // 8 g .. 8 g + 7, in three f16 pieces; per component also 1 / t_k, b_k for the lane's output dims, the constant
  f16x8 a1[KQ][2], a2[KQ][2], a3[KQ][2];
  float inv_t[KQ], zb[KQ][2], bl[KQ][2], cq[KQ];   // zb = -t b_hi (exact: t is a power of two)
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int k = wave + FW * q;
    const bool kv = k < K;
    const float* Ak = Ainv + ((int64_t)c * K + (kv ? k : 0)) * DP * DP;
    float v[2][8], amax = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int d = 8 * g + j;
        const float s = cscale[d];
        v[nt][j] = (kv && s > 0.f) ? Ak[(16 * nt + ln) * DP + d] / s : 0.f;
        amax = fmaxf(amax, fabsf(v[nt][j]));
      }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    const float t = fc_pow2_scale(amax);
    inv_t[q] = 1.f / t;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = v[nt][j] * t;
        // first piece on a grid relative to the largest entry (multiples of 1 below 2^10): the matrix instruction
        // cuts every addend off, toward zero, a few bits below the largest addend's last bit, so full-mantissa
        // pieces of small entries would lose with a sign in every row; their low bits go to the later pieces
        const _Float16 hi = fabsf(x) < 1024.f ? (_Float16)rintf(x) : (_Float16)x;
        const _Float16 lo = (_Float16)(x - (float)hi);
        a1[q][nt][j] = hi;
        a2[q][nt][j] = lo;
        a3[q][nt][j] = (_Float16)((x - (float)hi) - (float)lo);
      }
      zb[q][nt] = kv ? -t * bvec[((int64_t)c * K + k) * DP + 16 * nt + ln] : 0.f;
      bl[q][nt] = kv ? blo[((int64_t)c * K + k) * DP + 16 * nt + ln] : 0.f;
    }
    cq[q] = kv ? cst[c * K + k] : -INFINITY;
  }

  if (tid < FW * KQ) cks[tid] = tid < K ? cst[c * K + tid] : -INFINITY;   // visible after the barrier below

  f32x4 S2[KQ][2][2];
  float S1[KQ][2];
  float S0a = 0.f, S0b = 0.f;                 // sum of rho for components lane & 7 (and + FW) over rows 8 wave + (lane >> 3)
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      S1[q][a] = 0.f;
#pragma unroll
      for (int b = 0; b < 2; ++b) S2[q][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  double logp_acc = 0.0;                      // running total of the rows' log-densities minus row_level
  const float row_level = fullcov_row_level(cst + (int64_t)c * K, K, D);

  // staging: thread -> (row = tid / 8, columns 4 (tid % 8) .. + 3)
  const int srow = tid >> 3, sc4 = (tid & 7) * 4;
  float cm[4], cs[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { cm[e] = center[sc4 + e]; cs[e] = cscale[sc4 + e]; }
  const bool vec_ok = D == DP && (((uintptr_t)Y & 15) == 0);
  float yraw[4];
  auto gload = [&](int64_t tile) {
    const int64_t gr = tile * FT + srow;
    if (vec_ok && gr < n) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(Y + gr * DP + sc4);
#pragma unroll
      for (int e = 0; e < 4; ++e) yraw[e] = t4[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        yraw[e] = cm[e];
        if (gr < n && sc4 + e < D) yraw[e] = Y[gr * D + sc4 + e];
      }
    }
  };
  auto lstore = [&](int buf) {
    unsigned hi[2], lo[2];
    float q[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) q[e] = (yraw[e] - cm[e]) * cs[e];
    fc_split_pair(q[0], q[1], hi[0], lo[0]);
    fc_split_pair(q[2], q[3], hi[1], lo[1]);
    // 8-B piece (tid & 1) of 16-B chunk (tid & 7) >> 1 of the row
    const int off = srow * 64 + ((((tid & 7) >> 1) ^ swz(srow)) << 4) + 8 * (tid & 1);
    *reinterpret_cast<uint2*>(&ybuf[buf][0][off]) = make_uint2(hi[0], hi[1]);
    *reinterpret_cast<uint2*>(&ybuf[buf][1][off]) = make_uint2(lo[0], lo[1]);
  };

  // This workgroup's tiles: those of its row block that more than three components reach (fullcov_tilelists, the last
  // list); the others go to fullcov_single_kernel.
  const int2* seg = tlist + ((int64_t)c * gridDim.x + blockIdx.x) * cap;
  const int* so = toff + ((int64_t)c * gridDim.x + blockIdx.x) * kListOffs;
  const int li1 = so[kLists];
  int li = so[kLists - 1];
  // bit k of a tile's mask clear: component k is out of reach of every row of the tile (fullcov_bounds); the mask
  // comes with the list entry
  const bool tm = PRUNE && tmask != nullptr;
  unsigned mask_cur = 0xFFFFu, mask_nxt = 0xFFFFu;
  int64_t tile = 0, tile_nxt = 0;
  if (li < li1) {
    const int2 e = seg[li];
    tile = e.x;
    gload(tile);
    lstore(0);
    if (tm) mask_cur = (unsigned)e.y;
  }
  __syncthreads();
  int buf = 0;
  for (; li < li1; ++li, tile = tile_nxt) {
    const bool more = li + 1 < li1;
    if (more) {
      const int2 e = seg[li + 1];
      tile_nxt = e.x;
      gload(tile_nxt);
      if (tm) mask_nxt = (unsigned)e.y;
    }
    const unsigned tile_mask = __builtin_amdgcn_readfirstlane(mask_cur);
    mask_cur = mask_nxt;
    bool act[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) act[q] = !PRUNE || (wave + FW * q < K && ((tile_mask >> (wave + FW * q)) & 1u) != 0);
    const unsigned char* yh = ybuf[buf][0];
    const unsigned char* yl = ybuf[buf][1];
    // ---- GEMM1: u = A (y - mu) for the wave's components, rows 4 g + i of each 16-row tile in registers ----
    // Software-pipelined over the 4 KQ (row tile, component) groups: while the matrix pipe runs the 8 MFMAs of group
    // j + 1 (two dependent chains of four, interleaved), the vector ALU turns the accumulators of group j into u,
    // |u|^2 (4 fma per MFMA slot) and its sum over the 16 lanes of a row (4 DPP adds per slot).  Written as one MFMA
    // + one chunk of vector work per slot with a scheduling barrier after each, because left to itself the
    // compiler emits the eight MFMAs back to back and the 40 vector instructions after them.
    f32x4 u[GRAD ? KQ : 1][4][2];
    // the (row tile, component) groups of the components Q0 .. Q0 + NQ - 1 of this wave
    auto gemm1 = [&](auto q0_c, auto nq_c) {
      constexpr int Q0 = decltype(q0_c)::value, NQ = decltype(nq_c)::value;
      constexpr int NG = 4 * NQ;
      f16x8 fh[2], fl[2];                    // A operands of the row tile being multiplied / the next one
      f32x4 zc[2], zn[2];
      auto load_tile = [&](int rt, int slot) {
        const int off = (16 * rt + ln) * 64 + ((g ^ swz(16 * rt + ln)) << 4);
        fh[slot] = *reinterpret_cast<const f16x8*>(yh + off);
        fl[slot] = *reinterpret_cast<const f16x8*>(yl + off);
      };
      // MFMA m (0..7) of a group: chain nt = m & 1, product m >> 1 of (yh a1, yl a1, yh a2, yh a3), on top of -t b_hi
      auto mfma_step = [&](int q, int slot, int m, f32x4 (&z)[2]) {
        const int nt = m & 1, pr = m >> 1;
        const f16x8 av = pr <= 1 ? a1[q][nt] : (pr == 2 ? a2[q][nt] : a3[q][nt]);
        const f16x8 yv = pr == 1 ? fl[slot] : fh[slot];
        const f32x4 c0 = {zb[q][nt], zb[q][nt], zb[q][nt], zb[q][nt]};
        z[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yv, av, pr == 0 ? c0 : z[nt], 0, 0, 0);
      };
      float nrm[2][4];
      Row16Sum4 rs;
      load_tile(0, 0);
#pragma unroll
      for (int m = 0; m < 8; ++m) mfma_step(Q0, 0, m, zc);
#pragma unroll
      for (int j = 0; j < NG; ++j) {
        const int rt = j / NQ, q = Q0 + j % NQ;
        const int jn = j + 1, rtn = jn / NQ, qn = Q0 + jn % NQ;
        // The two waves of a SIMD (w and w + FW / 2) run the same stream between the same barriers; left alone the
        // older one wins every issue slot, reaches the barrier early and waits while the other finishes on its
        // own.  Issue priorities by phase, found by measurement (13.6 ms without any -> 12.9 ms; the eighteen other
        // patterns tried lie between 13.2 and 13.9): first half of this phase both waves high, second half and the
        // first half of the raw-sums phase below waves 4 - 7 high, its second half waves 0 - 3.  The condition has to
        // sit in a scalar register (on a per-lane condition both s_setprio are executed by every wave).
        if (GRAD && (j == 0 || j == NG / 2)) {
          if (j == 0 || !lead_wave) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
        }
        if (jn < NG && qn == Q0) load_tile(rtn, rtn & 1);
        // Value only: the row sums of group j - 1 ride in the odd slots of group j (no padding; measured faster
        // at four waves per SIMD).  With the gradient (two waves per SIMD) the padded steps in the group's own
        // slots measured faster.
        if constexpr (!GRAD) {
          // even slots: u and its squares for one row of group j; odd slots: one step of the row sums of group j - 1
          // (lane-crossing adds: the vector work of the slots between them covers their wait states)
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            if (jn < NG) mfma_step(qn, rtn & 1, m, zn);
            if ((m & 1) == 0) {
              const int i = m >> 1;
              const float u0 = fmaf(zc[0][i], inv_t[q], -bl[q][0]);
              const float u1 = fmaf(zc[1][i], inv_t[q], -bl[q][1]);
              if (GRAD) {
                u[GRAD ? q : 0][rt][0][i] = u0;
                u[GRAD ? q : 0][rt][1][i] = u1;
              }
              nrm[j & 1][i] = fmaf(u1, u1, u0 * u0);
            } else if (j > 0) {
              rs.template step<false>(m >> 1, nrm[(j - 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          // |u|^2 of row 16 rt + 4 g + b is now in the lanes of bank b
          if (j > 0 && (ln & 3) == 0)
            al[wave + FW * (Q0 + (j - 1) % NQ)][16 * ((j - 1) / NQ) + 4 * g + (ln >> 2)] = cq[Q0 + (j - 1) % NQ] - 0.5f * rs.d0;
        } else {
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            if (jn < NG) mfma_step(qn, rtn & 1, m, zn);
            if (m < 4) {
              const float u0 = fmaf(zc[0][m], inv_t[q], -bl[q][0]);
              const float u1 = fmaf(zc[1][m], inv_t[q], -bl[q][1]);
              if (GRAD) {
                u[GRAD ? q : 0][rt][0][m] = u0;
                u[GRAD ? q : 0][rt][1][m] = u1;
              }
              nrm[0][m] = fmaf(u1, u1, u0 * u0);
            } else {
              rs.template step<true>(m - 4, nrm[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          if ((ln & 3) == 0) al[wave + FW * q][16 * rt + 4 * g + (ln >> 2)] = cq[q] - 0.5f * rs.d0;
        }
        if (jn < NG) { zc[0] = zn[0]; zc[1] = zn[1]; }
      }
      if constexpr (!GRAD) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rs.template step<true>(k, nrm[(NG - 1) & 1]);
        if ((ln & 3) == 0)
          al[wave + FW * (Q0 + (NG - 1) % NQ)][16 * ((NG - 1) / NQ) + 4 * g + (ln >> 2)] = cq[Q0 + (NG - 1) % NQ] - 0.5f * rs.d0;
      }
    };
    // A component the tile's mask rules out is not evaluated: its log-densities enter the log-sum-exp as -inf (so
    // its responsibilities, flags and raw sums are exactly zero), the matrix and vector work of its groups is skipped
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    if constexpr (!PRUNE) {
      gemm1(I0{}, std::integral_constant<int, KQ>{});
    } else if constexpr (KQ == 1) {
      if (act[0]) gemm1(I0{}, I1{});
    } else {
      if (act[0] && act[1]) gemm1(I0{}, I2{});
      else if (act[0]) gemm1(I0{}, I1{});
      else if (act[1]) gemm1(I1{}, I1{});
    }
    if constexpr (PRUNE) {
#pragma unroll
      for (int q = 0; q < KQ; ++q)
        if (!act[q] && wave + FW * q < K) al[wave + FW * q][lane] = -INFINITY;
    }
    __syncthreads();
    {
      // log-sum-exp over the components and sqrt(rho_ik) = exp((a_ik - lse) / 2), shared out over the workgroup: wave w
      // takes rows 8 w .. 8 w + 7, lane = (component pair kp and kp + 8, row); maxima and sums cross the eight
      // neighbouring component lanes of a row with three DPP exchanges (a few cycles each; through ds_bpermute the
      // six dependent exchanges were a tenth of the tile's time, both waves of a SIMD sitting in them together).
      // Every (row, component) is exponentiated once per tile -- when every wave did all 64 rows for itself, the
      // K exponentials per lane were a seventh of the kernel's vector work.
      const int kp = lane & 7, rl = lane >> 3, row = 8 * wave + rl;
      const bool v0 = kp < K, v1 = KQ == 2 && kp + FW < K;
      const float a0 = v0 ? al[kp][row] : -INFINITY;
      const float a1 = v1 ? al[KQ == 2 ? kp + FW : kp][row] : -INFINITY;
      float m = fmaxf(a0, a1);
      m = fmaxf(m, lane_dpp<0xB1>(m));
      m = fmaxf(m, lane_dpp<0x4E>(m));
      m = fmaxf(m, lane_dpp<0x141>(m));
      float sm = __builtin_amdgcn_exp2f(1.44269504f * (a0 - m)) + __builtin_amdgcn_exp2f(1.44269504f * (a1 - m));
      sm += lane_dpp<0xB1>(sm);
      sm += lane_dpp<0x4E>(sm);
      sm += lane_dpp<0x141>(sm);
      const float lse = m + __logf(sm);
      const bool valid = tile * FT + row < n;
      if (kp == 0 && valid) logp_acc += (double)(lse - row_level);
      if (GRAD) {
        // a row whose |u|^2 = 2 (const - a) is beyond the f16 range of w (absurd parameters) is dropped from
        // the gradient sums instead of turning them into NaN; its log-density is already counted
        const float lsev = valid ? lse : INFINITY;
        float s0 = 0.f, s1 = 0.f;
        if (v0) {
          s0 = (a0 > cks[kp] - 1.5e9f ? 1.f : 0.f) * __builtin_amdgcn_exp2f(0.72134752f * (a0 - lsev));
          sq[kp][row] = s0;
          S0a = fmaf(s0, s0, S0a);
        }
        if (v1) {
          s1 = (a1 > cks[KQ == 2 ? kp + FW : kp] - 1.5e9f ? 1.f : 0.f) *
               __builtin_amdgcn_exp2f(0.72134752f * (a1 - lsev));
          sq[KQ == 2 ? kp + FW : kp][row] = s1;
          S0b = fmaf(s1, s1, S0b);
        }
        // Which components have, among this wave's eight rows, a responsibility the raw-sums phase can see: w =
        // sqrt(rho) u goes into f16 pieces, so sqrt(rho) < 2^-20 (rho < 2^-40) leaves nothing of a row in S2 and
        // less than 2^-40 |u| of it in S1.  A 32-row half of the tile without such a row for a component skips
        // that component's share of the phase -- most of it when neighbouring rows belong to the same few
        // components (rows stored cluster by cluster; any order is correct).  NaN counts as worth multiplying.
        const unsigned long long m0 = __builtin_amdgcn_ballot_w64(!(s0 < 9.5367431640625e-07f));
        const unsigned long long m1 = __builtin_amdgcn_ballot_w64(!(s1 < 9.5367431640625e-07f));
        if (lane < FW) {
          actf[0][lane][wave] = ((m0 >> lane) & 0x0101010101010101ull) != 0 ? 1 : 0;
          actf[1][lane][wave] = ((m1 >> lane) & 0x0101010101010101ull) != 0 ? 1 : 0;
        }
      }
    }
    if (GRAD) __syncthreads();
    // ---- responsibilities and raw sums: w = sqrt(rho) u in two f16 pieces is both operand of S2 += W^T W ----
#pragma unroll
    for (int q = 0; q < (GRAD ? KQ : 0); ++q) {
      const int k = wave + FW * q;
      if (lead_wave == (q != 0)) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
      if (k < K) {
        // rows 0 .. 31 were looked at by waves 0 .. 3 (bytes 0 .. 3), rows 32 .. 63 by waves 4 .. 7
        const uint2 seen = *reinterpret_cast<const uint2*>(&actf[q][wave][0]);
#pragma unroll
        for (int m = 0; m < 2; ++m) {        // 32-row pairs: k index j <-> row 32 m + 16 (j >> 2) + 4 g + (j & 3)
          if (__builtin_amdgcn_readfirstlane(m == 0 ? seen.x : seen.y) == 0) continue;
          float sr[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int row = 32 * m + 16 * (j >> 2) + 4 * g + (j & 3);
            sr[j] = sq[k][row];
          }
          f16x8 wh[2], wl[2];
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            unsigned hw[4], lw[4];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              const float w0 = sr[j] * u[q][2 * m + (j >> 2)][nt][j & 3];
              const float w1 = sr[j + 1] * u[q][2 * m + (j >> 2)][nt][(j & 3) + 1];
              S1[q][nt] = fmaf(sr[j], w0, S1[q][nt]);
              S1[q][nt] = fmaf(sr[j + 1], w1, S1[q][nt]);
              fc_split_pair(w0, w1, hw[j >> 1], lw[j >> 1]);
            }
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            wh[nt] = __builtin_bit_cast(f16x8, (u32x4){hw[0], hw[1], hw[2], hw[3]});
            wl[nt] = __builtin_bit_cast(f16x8, (u32x4){lw[0], lw[1], lw[2], lw[3]});
          }
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = a; b < 2; ++b) {      // S2 is symmetric: tile (1, 0) is the transpose of (0, 1)
              S2[q][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[a], wh[b], S2[q][a][b], 0, 0, 0);
              S2[q][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[a], wh[b], S2[q][a][b], 0, 0, 0);
              S2[q][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[a], wl[b], S2[q][a][b], 0, 0, 0);
            }
        }
      }
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- partials of this (row block, chain): [K][1 + DP + DP*DP] then the log-prob ----
  // S2 accumulator: lane holds rows (first index) 16 a + 4 g + i, column 16 b + ln
  constexpr int Q = 1 + DP + DP * DP;
  float* out = partials + ((int64_t)blockIdx.x * C + c) * ((int64_t)K * Q + 1);
#pragma unroll
  for (int q = 0; q < (GRAD ? KQ : 0); ++q) {
    const int k = wave + FW * q;
    if (k < K) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float s1 = S1[q][nt];
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (g == 0) out[(int64_t)k * Q + 1 + 16 * nt + ln] = s1;
#pragma unroll
        for (int b = nt; b < 2; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            out[(int64_t)k * Q + 1 + DP + (16 * nt + 4 * g + i) * DP + 16 * b + ln] = S2[q][nt][b][i];
            if (b != nt) out[(int64_t)k * Q + 1 + DP + (16 * b + ln) * DP + 16 * nt + 4 * g + i] = S2[q][nt][b][i];
          }
      }
    }
  }
  if ((lane & 7) == 0) lpw[8 * wave + (lane >> 3)] = (float)logp_acc;   // the lanes that summed rows 8 w + (0..7)
  if (GRAD) {
    // S0: add the eight row lanes of a component, then the waves (the tile buffers are free by now)
    float* s0w = reinterpret_cast<float*>(&al[0][0]);          // [FW][2 * FW]
#pragma unroll
    for (int off = 8; off <= 32; off <<= 1) {
      S0a += __shfl_xor(S0a, off, 64);
      S0b += __shfl_xor(S0b, off, 64);
    }
    __syncthreads();
    if (lane < 8) {
      s0w[wave * 2 * FW + lane] = S0a;
      s0w[wave * 2 * FW + FW + lane] = S0b;
    }
  }
  __syncthreads();
  if (GRAD && tid < K) {
    const float* s0w = reinterpret_cast<const float*>(&al[0][0]);
    float s = 0.f;
    for (int w = 0; w < FW; ++w) s += s0w[w * 2 * FW + tid];
    out[(int64_t)tid * Q] = s;
  }
  if (tid == 0) {
    float s = 0.f;
    for (int i = 0; i < FT; ++i) s += lpw[i];
    out[(int64_t)K * Q] = s;
  }
}


// The rows as fullcov_single_kernel multiplies them: (y - centre) scale in two f16 pieces, [tile rows][DP] each, rows
// past the end and columns past D zero.  The split does not depend on the chain, and in that kernel -- one component,
// no log-sum-exp -- it was a third of all vector instructions when every (chain, tile) redid it; one pass per call
// (skipped when no tile is on a single-component list) does it once.
__global__ __launch_bounds__(256) void fullcov_planes(const float* __restrict__ Y, int64_t n, int D, int64_t ntiles,
                                                      const float* __restrict__ center, const float* __restrict__ cscale,
                                                      const int* __restrict__ any_single,
                                                      unsigned char* __restrict__ phi, unsigned char* __restrict__ plo) {
  if (*any_single == 0) return;
  const int g = threadIdx.x & 3;
  float cm[8], cs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { cm[e] = center[8 * g + e]; cs[e] = cscale[8 * g + e]; }
  const bool vec_ok = D == DP && (((uintptr_t)Y & 15) == 0);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  for (int64_t row = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2); row < ntiles * FT; row += (int64_t)gridDim.x * 64) {
    float y[8];
    if (vec_ok && row < n) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(Y + row * DP + 8 * g);
      const f32x4 b = *reinterpret_cast<const f32x4*>(Y + row * DP + 8 * g + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { y[e] = a[e]; y[4 + e] = b[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool in = row < n && 8 * g + e < D;
        const float v = Y[in ? row * D + 8 * g + e : 0];       // (a tile exists, so Y[0] does)
        y[e] = in ? v : cm[e];
      }
    }
    unsigned hw[4], lw[4];
#pragma unroll
    for (int e = 0; e < 8; e += 2) fc_split_pair((y[e] - cm[e]) * cs[e], (y[e + 1] - cm[e + 1]) * cs[e + 1], hw[e >> 1], lw[e >> 1]);
    *reinterpret_cast<u32x4*>(phi + row * 64 + 16 * g) = (u32x4){hw[0], hw[1], hw[2], hw[3]};
    *reinterpret_cast<u32x4*>(plo + row * 64 + 16 * g) = (u32x4){lw[0], lw[1], lw[2], lw[3]};
  }
}

// A''_k = t_k A_k diag(1 / column scale) as the B operand of the residual product, in three f16 pieces (exact), per
// chain and component, in the lane layout fullcov_single_kernel loads it with: apc[((chain K + k) 3 + piece) 2 + nt][lane]
// (lane (ln, g): row 16 nt + ln of A, columns 8 g .. 8 g + 7); at[chain K + k] = the power of two t_k.  Computed once
// per call: that kernel needs the pieces of its own component once per wave and those of a tile's other components
// once per tile.
__global__ __launch_bounds__(64) void fullcov_apieces(const float* __restrict__ Ainv, const float* __restrict__ cscale,
                                                      f16x8* __restrict__ apc, float* __restrict__ at) {
  const int lane = threadIdx.x, ln = lane & 15, g = lane >> 4;
  const float* Ak = Ainv + (int64_t)blockIdx.x * DP * DP;
  float v[2][8], amax = 0.f;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = 8 * g + j;
      const float sc = cscale[d];
      v[nt][j] = sc > 0.f ? Ak[(16 * nt + ln) * DP + d] / sc : 0.f;
      amax = fmaxf(amax, fabsf(v[nt][j]));
    }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
  const float t = fc_pow2_scale(amax);
  if (lane == 0) at[blockIdx.x] = t;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    f16x8 a1, a2, a3;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = v[nt][j] * t;
      const _Float16 hi = fabsf(x) < 1024.f ? (_Float16)rintf(x) : (_Float16)x;
      const _Float16 lo = (_Float16)(x - (float)hi);
      a1[j] = hi;
      a2[j] = lo;
      a3[j] = (_Float16)((x - (float)hi) - (float)lo);
    }
    apc[(((int64_t)blockIdx.x * 3 + 0) * 2 + nt) * 64 + lane] = a1;
    apc[(((int64_t)blockIdx.x * 3 + 1) * 2 + nt) * 64 + lane] = a2;
    apc[(((int64_t)blockIdx.x * 3 + 2) * 2 + nt) * 64 + lane] = a3;
  }
}

// The tiles ONE component reaches, and those two or three reach.  On the first kind every responsibility is 1 and a
// row's log-density is that component's: no log-sum-exp, nothing to exchange between waves, no LDS staging and no
// barrier.  Every wave works through its own lists of tiles (fullcov_tilelists) by itself, one component after the
// other: it loads the f16 pieces of a tile's rows (fullcov_planes) straight into the A-operand layout of the residual
// product (lane = row 16 rt + ln, dims 8 g .. 8 g + 7; the next tile's rows are in flight meanwhile) and runs the same
// two products as fullcov_f16_kernel: residuals (four f16 products), |u|^2 by DPP row sums, then S0 += 1, S1 += u,
// S2 += u u^T.  A tile of the second kind stands in the list of each of its components: the wave first runs the
// residual product for the OTHER components of the tile (their operand pieces come from fullcov_apieces) to get the
// rows' log-sum-exp, then its own, and adds its component's sums with the responsibilities as weights (S0 += rho,
// S1 += rho u, S2 += rho u u^T); the wave of the tile's first component also adds the rows' log-densities.  That is
// c residual products per component instead of one shared by all -- and no wave of eight waits for the two or three
// that have a component on the tile, which is what makes such tiles four times as expensive in fullcov_f16_kernel.
// Rows past the end or beyond the f16 range of u (absurd parameters) get weight 0 through a per-wave LDS line.
// Wave w owns components (w % K8) and (w % K8) + FW, K8 = FW / R; with R > 1 (K <= 4) wave w is replica w / K8 and
// writes the partial sums of block R * blockIdx.x + replica.
template <bool GRAD, bool FEW>
__global__ __launch_bounds__(64 * FW) __attribute__((amdgpu_waves_per_eu(2, GRAD ? 2 : 4))) void fullcov_single_kernel(
    int64_t n, int D, int K, const float* __restrict__ bvec, const float* __restrict__ blo,
    const float* __restrict__ cst, const f16x8* __restrict__ apc, const float* __restrict__ at, int C,
    float* __restrict__ partials, const int2* __restrict__ tlist, const int* __restrict__ toff, int cap,
    const unsigned char* __restrict__ phi, const unsigned char* __restrict__ plo) {
  __shared__ float sqw[FW][FT];
  __shared__ double lpw[FW];
  // (the wave number in a scalar register: the list entries are then scalar loads, which the waits for the rows'
  // vector loads do not see -- as a vector load the next entry drew a vmcnt(0) right behind the prefetch of the rows)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, g = lane >> 4;
  const int c = blockIdx.y;
  const int R = fullcov_replicas(K), K8 = FW / R, kbase = wave % K8, rep = wave / K8;

  constexpr int Q = 1 + DP + DP * DP;
  float* out = partials + (((int64_t)blockIdx.x * R + rep) * C + c) * ((int64_t)K * Q + 1);
  const int2* seg = tlist + ((int64_t)c * gridDim.x + blockIdx.x) * cap;
  const int* so = toff + ((int64_t)c * gridDim.x + blockIdx.x) * kListOffs;
  // this instantiation's part of list j: the tiles component j alone reaches (FEW = false: the kernel writes the
  // partial sums) or the shared tiles (FEW = true: launched after the other, it adds to them -- and has nothing to do
  // when no list of the (chain, row block) has a second part)
  auto first_entry = [&](int j) { return FEW ? so[kFewOff + j] : so[j]; };
  auto end_entry = [&](int j) { return FEW ? so[j + 1] : so[kFewOff + j]; };
  if constexpr (FEW) {
    bool any = false;
    for (int q = 0; q < 2; ++q) any = any || first_entry(wave + FW * q) < end_entry(wave + FW * q);
    if (!__syncthreads_or(any)) return;
  }
  double logp_acc = 0.0;
  const float row_level = fullcov_row_level(cst + (int64_t)c * K, K, D);
  f16x8 rawh[4], rawl[4];
  auto gload = [&](int64_t tile) {          // rows 16 rt + ln of the tile, dims 8 g .. 8 g + 7, both pieces
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int64_t off = (tile * FT + 16 * rt + ln) * 64 + 16 * g;
      rawh[rt] = *reinterpret_cast<const f16x8*>(phi + off);
      rawl[rt] = *reinterpret_cast<const f16x8*>(plo + off);
    }
  };
  // operand pieces and constants of component kk of this chain
  struct Comp {
    f16x8 a1[2], a2[2], a3[2];
    float inv_t, zb[2], bl[2], cq;
  };
  auto load_comp = [&](int kk, Comp& o) {
    const int64_t ck = (int64_t)c * K + kk;
    const float t = at[ck];
    o.inv_t = 1.f / t;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      o.a1[nt] = apc[((ck * 3 + 0) * 2 + nt) * 64 + lane];
      o.a2[nt] = apc[((ck * 3 + 1) * 2 + nt) * 64 + lane];
      o.a3[nt] = apc[((ck * 3 + 2) * 2 + nt) * 64 + lane];
      o.zb[nt] = -t * bvec[ck * DP + 16 * nt + ln];
      o.bl[nt] = blo[ck * DP + 16 * nt + ln];
    }
    o.cq = cst[ck];
  };
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  for (int q = 0; q < 2; ++q) {
    const int k = kbase + FW * q;
    if (k >= K) break;
    Comp own;
    load_comp(k, own);
    f32x4 S2[2][2];
    float S1[2] = {0.f, 0.f}, S0 = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) S2[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int li = first_entry(wave + FW * q);
    const int li1 = end_entry(wave + FW * q);
    if (FEW && li >= li1) continue;          // nothing to add for this component
    int2 ent = make_int2(0, 0), ent_after = make_int2(0, 0);   // ent_after: list entry li + 2, read a whole tile ahead
    if (li < li1) {
      ent = seg[li];
      gload(ent.x);
    }
    if (li + 1 < li1) ent_after = seg[li + 1];
    for (; li < li1; ++li) {
      // the tile's rows (A operands); the staging registers are then free for the next tile's rows
      f16x8 fh[4], fl[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) { fh[rt] = rawh[rt]; fl[rt] = rawl[rt]; }
      const int64_t tile_cur = ent.x;
      const unsigned tile_mask = (unsigned)__builtin_amdgcn_readfirstlane(ent.y);
      if (li + 1 < li1) {
        ent = ent_after;
        gload(ent.x);
      }
      if (li + 2 < li1) ent_after = seg[li + 2];
      f32x4 u[GRAD ? 4 : 1][2];
      // residual product of the tile with one component: arow[rt] = log-density of row 16 rt + 4 g + (ln >> 2) under
      // it (without the weight's share of the log-sum-exp: cq holds the weight); KEEP: the residuals stay in u
      auto resid = [&](auto keep, const Comp& o, float (&arow)[4]) {
        constexpr bool KEEP = decltype(keep)::value;
        f32x4 zc[2], zn[2];
        // MFMA m (0..7) of a row tile: chain nt = m & 1, product m >> 1 of (yh a1, yl a1, yh a2, yh a3), on top of -t b_hi
        auto mfma_step = [&](int rt, int m, f32x4 (&z)[2]) {
          const int nt = m & 1, pr = m >> 1;
          const f16x8 av = pr <= 1 ? o.a1[nt] : (pr == 2 ? o.a2[nt] : o.a3[nt]);
          const f16x8 yv = pr == 1 ? fl[rt] : fh[rt];
          const f32x4 c0 = {o.zb[nt], o.zb[nt], o.zb[nt], o.zb[nt]};
          z[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yv, av, pr == 0 ? c0 : z[nt], 0, 0, 0);
        };
        float nrm[4];
        Row16Sum4 rs;
#pragma unroll
        for (int m = 0; m < 8; ++m) mfma_step(0, m, zc);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            if (rt + 1 < 4) mfma_step(rt + 1, m, zn);
            if (m < 4) {
              const float u0 = fmaf(zc[0][m], o.inv_t, -o.bl[0]);
              const float u1 = fmaf(zc[1][m], o.inv_t, -o.bl[1]);
              if constexpr (KEEP) {
                u[KEEP ? rt : 0][0][m] = u0;
                u[KEEP ? rt : 0][1][m] = u1;
              }
              nrm[m] = fmaf(u1, u1, u0 * u0);
            } else {
              rs.template step<true>(m - 4, nrm);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          arow[rt] = o.cq - 0.5f * rs.d0;
          if (rt + 1 < 4) { zc[0] = zn[0]; zc[1] = zn[1]; }
        }
      };
      // the tile's other components: running log-sum-exp of their log-densities per row
      const unsigned others = tile_mask & ~(1u << k);
      float om[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, os[4] = {0.f, 0.f, 0.f, 0.f};
      for (unsigned rest = FEW ? others : 0u; rest != 0; rest &= rest - 1) {      // (the same in every lane)
        Comp oth;
        load_comp(__builtin_ctz(rest), oth);
        float ao[4];
        resid(std::false_type{}, oth, ao);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const float mn = fmaxf(om[rt], ao[rt]);
          const float mm = mn > -INFINITY ? mn : 0.f;
          os[rt] = os[rt] * __expf(om[rt] - mm) + __expf(ao[rt] - mm);      // (a NaN log-density stays in the sum)
          om[rt] = mn;
        }
      }
      float arow[4];
      resid(std::integral_constant<bool, GRAD>{}, own, arow);
      // log-densities (one of the four lanes that hold a row's value adds it; on a shared tile only the wave of its
      // first component) and the rows' weights (the square roots of the responsibilities)
      const bool adds_logp = (tile_mask & ((1u << k) - 1u)) == 0;
      bool ok = true;
      float sv[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const bool valid = tile_cur * FT + 16 * rt + 4 * g + (ln >> 2) < n;
        const float a = arow[rt];
        float lp = a, wgt = 1.f;
        if constexpr (FEW) {
          const float mn = fmaxf(om[rt], a);
          const float mm = mn > -INFINITY ? mn : 0.f;
          const float st = os[rt] * __expf(om[rt] - mm) + __expf(a - mm);
          lp = mm + __logf(st);
          wgt = a > -INFINITY ? sqrtf(__expf(a - lp)) : 0.f;
          if (a != a) wgt = a;
        }
        if (valid && adds_logp && (ln & 3) == 0) logp_acc += (double)(lp - row_level);
        sv[rt] = valid ? (a > own.cq - 1.5e9f ? wgt : (a != a ? a : 0.f)) : 0.f;
        ok = ok && sv[rt] == 1.f;
        if (GRAD && (ln & 3) == 0) S0 = fmaf(sv[rt], sv[rt], S0);
      }
      if constexpr (GRAD) {
        const bool all_ok = __builtin_amdgcn_ballot_w64(ok) == ~0ull;
        if (!all_ok) {
#pragma unroll
          for (int rt = 0; rt < 4; ++rt)
            if ((ln & 3) == 0) sqw[wave][16 * rt + 4 * g + (ln >> 2)] = sv[rt];
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {        // 32-row pairs: k index j <-> row 32 m + 16 (j >> 2) + 4 g + (j & 3)
          f16x8 wh[2], wl[2];
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            unsigned hw[4], lw[4];
            if (all_ok) {                    // every row counts in full (all but the last tile, usually): w = u
#pragma unroll
              for (int j = 0; j < 8; j += 2) {
                const float w0 = u[2 * m + (j >> 2)][nt][j & 3];
                const float w1 = u[2 * m + (j >> 2)][nt][(j & 3) + 1];
                S1[nt] += w0 + w1;
                fc_split_pair(w0, w1, hw[j >> 1], lw[j >> 1]);
              }
            } else {
#pragma unroll
              for (int j = 0; j < 8; j += 2) {
                const float sr0 = sqw[wave][32 * m + 16 * (j >> 2) + 4 * g + (j & 3)];
                const float sr1 = sqw[wave][32 * m + 16 * (j >> 2) + 4 * g + (j & 3) + 1];
                const float w0 = sr0 * u[2 * m + (j >> 2)][nt][j & 3];
                const float w1 = sr1 * u[2 * m + (j >> 2)][nt][(j & 3) + 1];
                S1[nt] = fmaf(sr0, w0, S1[nt]);
                S1[nt] = fmaf(sr1, w1, S1[nt]);
                fc_split_pair(w0, w1, hw[j >> 1], lw[j >> 1]);
              }
            }
            wh[nt] = __builtin_bit_cast(f16x8, (u32x4){hw[0], hw[1], hw[2], hw[3]});
            wl[nt] = __builtin_bit_cast(f16x8, (u32x4){lw[0], lw[1], lw[2], lw[3]});
          }
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = a; b < 2; ++b) {
              S2[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[a], wh[b], S2[a][b], 0, 0, 0);
              S2[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[a], wh[b], S2[a][b], 0, 0, 0);
              S2[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[a], wl[b], S2[a][b], 0, 0, 0);
            }
        }
      }
}

    // ---- partials of (row block, replica, chain), component k: the layout of fullcov_f16_kernel ----
    if constexpr (GRAD) {
      float s0 = S0;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) s0 += __shfl_xor(s0, off, 64);
      auto put = [&](int64_t idx, float v) { out[idx] = FEW ? out[idx] + v : v; };
      if (lane == 0) put((int64_t)k * Q, s0);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float s1 = S1[nt];
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (g == 0) put((int64_t)k * Q + 1 + 16 * nt + ln, s1);
#pragma unroll
        for (int b = nt; b < 2; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            put((int64_t)k * Q + 1 + DP + (16 * nt + 4 * g + i) * DP + 16 * b + ln, S2[nt][b][i]);
            if (b != nt) put((int64_t)k * Q + 1 + DP + (16 * b + ln) * DP + 16 * nt + 4 * g + i, S2[nt][b][i]);
          }
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) logp_acc += __shfl_xor(logp_acc, off, 64);
  if (lane == 0) lpw[wave] = logp_acc;
  __syncthreads();
  if (lane == 0 && kbase == 0) {
    double t = 0.0;
    for (int w = 0; w < K8; ++w) t += lpw[rep * K8 + w];
    out[(int64_t)K * Q] = FEW ? (float)((double)out[(int64_t)K * Q] + t) : (float)t;
  }
}

// one block of 8 x DP threads per (chain, component): thread (j, ig) owns column j of the rows i = ig, ig + 8, ...
__global__ __launch_bounds__(8 * DP) void fullcov_finalize(const float* __restrict__ partials, int nb, int64_t n,
                                                           int D, int K, const float* __restrict__ theta,
                                                           const float* __restrict__ Ainv,
                                                           const float* __restrict__ cst, int C,
                                                           float* __restrict__ logp, float* __restrict__ grad,
                                                           const double* __restrict__ ref,
                                                           const int* __restrict__ toff, int nb_dense) {
  constexpr int Q = 1 + DP + DP * DP, RG = DP / 8;
  const int c = blockIdx.x / K, k = blockIdx.x % K, j = threadIdx.x & (DP - 1), ig = threadIdx.x / DP;
  const int P = K + K * D + K * D * D;
  const int64_t stride = (int64_t)K * Q + 1;
  __shared__ double S2[DP][DP + 1], S1[DP], S0s, A[DP][DP + 1];
  double s1 = 0.0, s0 = 0.0, lp = 0.0;
  double col[RG];
  for (int r = 0; r < RG; ++r) col[r] = 0.0;
  // a row block without tiles for the many-component kernel wrote nothing: its flags first, all at once (inside the
  // loop each was a pair of dependent loads in front of the block's partial sums)
  __shared__ unsigned char skip_dense[512];        // (fullcov_blocks gives at most 512 row blocks)
  const int nsk = toff ? (nb_dense < 512 ? nb_dense : 512) : 0;
  for (int b = threadIdx.x; b < nsk; b += blockDim.x) {
    const int* so = toff + ((int64_t)c * nb_dense + b) * kListOffs;
    skip_dense[b] = so[kLists - 1] == so[kLists] ? 1 : 0;
  }
  __syncthreads();
  auto skipped = [&](int b) { return b < nsk && skip_dense[b] != 0; };
  auto add_block = [&](int b) {
    const float* p = partials + ((int64_t)b * C + c) * stride + (int64_t)k * Q;
    if (grad) {                     // a value-only launch wrote the log-density partial only
      for (int r = 0; r < RG; ++r) col[r] += (double)p[1 + DP + (ig + 8 * r) * DP + j];
      if (ig == 0) s1 += (double)p[1 + j];
    }
    if (threadIdx.x == 0) {
      if (grad) s0 += (double)p[0];
      if (k == 0) lp += (double)partials[((int64_t)b * C + c) * stride + (int64_t)K * Q];
    }
  };
  if (grad) {
    // four blocks' loads in flight at a time; the sums keep the order b = 0, 1, 2, ...
    int b = 0;
    for (; b + 4 <= nb; b += 4) {
      float v[4][RG], v1[4], v0[4], vl[4];
      // every load unconditional (the addresses are inside the workspace whatever was written there), the selection
      // afterwards: conditional loads would each sit in a branch of their own with a wait behind it
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* p = partials + ((int64_t)(b + u) * C + c) * stride + (int64_t)k * Q;
#pragma unroll
        for (int r = 0; r < RG; ++r) v[u][r] = p[1 + DP + (ig + 8 * r) * DP + j];
        v1[u] = p[1 + j];
        v0[u] = p[0];
        vl[u] = partials[((int64_t)(b + u) * C + c) * stride + (int64_t)K * Q];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool on = !skipped(b + u);
#pragma unroll
        for (int r = 0; r < RG; ++r) v[u][r] = on ? v[u][r] : 0.f;
        v1[u] = (on && ig == 0) ? v1[u] : 0.f;
        v0[u] = (on && threadIdx.x == 0) ? v0[u] : 0.f;
        vl[u] = (on && threadIdx.x == 0 && k == 0) ? vl[u] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int r = 0; r < RG; ++r) col[r] += (double)v[u][r];
        s1 += (double)v1[u];
        s0 += (double)v0[u];
        lp += (double)vl[u];
      }
    }
    for (; b < nb; ++b)
      if (!skipped(b)) add_block(b);
  } else {
    for (int b = 0; b < nb; ++b)
      if (!skipped(b)) add_block(b);
  }
  // the f16 kernel sums the rows' log-densities around a typical row's level (fullcov_row_level): back in double
  if (cst && threadIdx.x == 0 && k == 0) lp += (double)n * (double)fullcov_row_level(cst + (int64_t)c * K, K, D);
  for (int r = 0; r < RG; ++r) {
    const int i = ig + 8 * r;
    S2[i][j] = col[r];
    A[i][j] = (double)Ainv[((int64_t)c * K + k) * DP * DP + i * DP + j];
  }
  if (ig == 0) S1[j] = s1;
  if (threadIdx.x == 0) S0s = s0;
  __syncthreads();
  if (threadIdx.x == 0 && k == 0) logp[c] = (float)(lp - (ref ? *ref : 0.0));
  if (!grad) return;
  const float* th = theta + (int64_t)c * P;
  float* gc = grad + (int64_t)c * P;
  if (threadIdx.x == 0) {
    double wsum = 0.0;
    for (int m = 0; m < K; ++m) wsum += (double)th[m];
    gc[k] = (float)(S0s / (double)th[k] - (double)n / wsum);
  }
  if (j < D) {
    // d/dmu_j = sum_m A[m][j] S1[m];  d/dL[i][j] (i >= j) = sum_m A[m][i] S2[m][j] - S0 [i == j] / L_jj
    if (ig == 0) {
      double gm = 0.0;
      for (int m = 0; m < D; ++m) gm += A[m][j] * S1[m];
      gc[K + k * D + j] = (float)gm;
    }
    const float* L = th + K + K * D + k * D * D;
    for (int i = ig; i < D; i += 8) {
      double v = 0.0;
      if (i >= j) {
        for (int m = 0; m < D; ++m) v += A[m][i] * S2[m][j];
        if (i == j) v -= S0s / (double)L[j * D + j];
      }
      gc[K + K * D + k * D * D + i * D + j] = (float)v;
    }
  }
}

int fullcov_blocks(int64_t n, int c, bool grad = false) {
  int nb = bvi_ceil_div(n, 4 * TR);
  // workgroups per CU over all chains: 2 for the value-only kernels (two fit), 1 with the gradient (one fits: a second
  // round would pay the prologue and the partial sums twice); sizes are taken for the larger of the two
  const int per = grad ? 256 : 512;
  const int cap = c >= per ? 1 : per / c;
  if (nb > cap) nb = cap;
  return nb < 1 ? 1 : nb;
}

}  // namespace

constexpr int kStatBlocks = 2048;
// floats of workspace in front of everything else: column statistics of Y (doubles first: 8-B aligned)
constexpr size_t kStatFloats = 2 * (size_t)kStatBlocks * DP + 2 * (size_t)kStatBlocks * DP + 2 * DP;

// tiles per row block, at most
int fullcov_list_cap(int64_t n, int nb) {      // entries: a tile two or three components reach stands in a list of each
  const int64_t ntiles = (n + FT - 1) / FT;
  return kFewMax * ((int)((ntiles + nb - 1) / nb) + 1);
}

// floats for the far-component test and the tile lists: per-tile centre + radius, |A_k|_2 bounds, the tiles'
// component masks (16 bits each), the lists (one int per tile and (chain, block) slot) and their offsets
size_t fullcov_tile_floats(int64_t n, int c, int k) {
  const size_t ntiles = (size_t)((n + FT - 1) / FT);
  const size_t nb = (size_t)fullcov_blocks(n, c), nbg = (size_t)fullcov_blocks(n, c, true);
  // (an entry is two ints)
  const size_t lists = (size_t)c * nb * (2 * (size_t)fullcov_list_cap(n, (int)nb) + kListOffs);
  const size_t listsg = (size_t)c * nbg * (2 * (size_t)fullcov_list_cap(n, (int)nbg) + kListOffs);
  const size_t ngroups = (ntiles + 63) / 64;
  // + the flag "some tile is on a single-component list" and the two f16 planes of the rows (16-B aligned)
  return (ntiles + ngroups) * (DP + 1) + 3 * (size_t)c * k + 2 * (size_t)c * ngroups * k + 2 +
         ((size_t)c * ntiles + 1) / 2 + 4 +
         (lists > listsg ? lists : listsg) + 8 + ntiles * FT * DP;
}

size_t bvi_ws_gmm_fullcov(int64_t n, int c, int d, int k) {
  (void)d;
  // A, b (two floats), constants; t and the f16 pieces of A'' (6 KB per chain and component, 16-B aligned)
  const size_t prep = (size_t)c * k * (DP * DP + 2 * DP + 1) + (size_t)c * k * (1 + 3 * 2 * 64 * 4) + 4;
  // partial sums: one set per row block from the kernel of the tiles several components reach, R sets per row
  // block from the kernel of the tiles one component reaches
  const size_t part = (size_t)fullcov_blocks(n, c) * (1 + fullcov_replicas(k)) * c * ((size_t)k * (1 + DP + DP * DP) + 1);
  return (kStatFloats + prep + part + fullcov_tile_floats(n, c, k)) * sizeof(float);
}

// What of a call depends on the rows alone, computed once by bvi_launch_gmm_fullcov_pack: column centres and scales
// (DP floats each), the tiles' and the tile groups' centres and radii, the two f16 planes of the rows (16-B aligned).
namespace {
struct PackedLayout {
  size_t center, cscale, tstat, gstat, phi, plo, total;
};
PackedLayout packed_layout(int64_t n) {
  const size_t ntiles = (size_t)((n + FT - 1) / FT), ngroups = (ntiles + 63) / 64;
  PackedLayout L;
  size_t o = 0;
  L.center = o; o += DP * sizeof(float);
  L.cscale = o; o += DP * sizeof(float);
  L.tstat = o;  o += ntiles * (DP + 1) * sizeof(float);
  L.gstat = o;  o += ngroups * (DP + 1) * sizeof(float);
  o = (o + 255) & ~(size_t)255;
  L.phi = o;    o += ntiles * FT * DP * 2;
  L.plo = o;    o += ntiles * FT * DP * 2;
  L.total = o + 256;
  return L;
}
}  // namespace

size_t bvi_gmm_fullcov_packed_size(int64_t n, int d) {
  (void)d;
  return packed_layout(n).total;
}

size_t bvi_gmm_fullcov_pack_scratch(int d) {
  (void)d;
  return kStatFloats * sizeof(float) + 256;
}

int bvi_launch_gmm_fullcov_pack(const float* Y, int64_t n, int d, void* packed, void* scratch, hipStream_t s) {
  if (d < 1 || d > DP) return BVI_ERR_UNSUPPORTED;
  if (((uintptr_t)scratch & 7) != 0 || ((uintptr_t)packed & 15) != 0) return BVI_ERR_ARG;
  const int64_t ntiles = (n + FT - 1) / FT, ngroups = (ntiles + 63) / 64;
  const PackedLayout L = packed_layout(n);
  unsigned char* pk = static_cast<unsigned char*>(packed);
  float* ws = static_cast<float*>(scratch);
  double* psum = (double*)ws;
  float* pmin = ws + 2 * (size_t)kStatBlocks * DP;
  float* pmax = pmin + (size_t)kStatBlocks * DP;
  int* flag = reinterpret_cast<int*>(ws + kStatFloats);
  float* center = reinterpret_cast<float*>(pk + L.center);
  float* cscale = reinterpret_cast<float*>(pk + L.cscale);
  float* tstat = reinterpret_cast<float*>(pk + L.tstat);
  float* gstat = reinterpret_cast<float*>(pk + L.gstat);
  const int sblocks = (int)(ntiles / 4 < 1 ? 1 : (ntiles / 4 > kStatBlocks ? kStatBlocks : ntiles / 4));
  hipError_t me = hipMemsetAsync(flag, 1, sizeof(int), s);            // "the planes are wanted"
  if (me != hipSuccess) return (int)me;
  hipLaunchKernelGGL(fullcov_tilestats, dim3(sblocks), dim3(256), 0, s, Y, n, d, ntiles, tstat, psum, pmin, pmax);
  hipLaunchKernelGGL(fullcov_colstats_final, dim3(1), dim3(1024), 0, s, (const double*)psum, (const float*)pmin,
                     (const float*)pmax, sblocks, n, d, center, cscale);
  if (ntiles > 0) {
    hipLaunchKernelGGL(fullcov_groupstats, dim3((unsigned)ngroups), dim3(64), 0, s, (const float*)tstat, ntiles, gstat);
    const int pblocks = (int)(ntiles > 4096 ? 4096 : ntiles);
    hipLaunchKernelGGL(fullcov_planes, dim3(pblocks), dim3(256), 0, s, Y, n, d, ntiles, (const float*)center,
                       (const float*)cscale, (const int*)flag, pk + L.phi, pk + L.plo);
  }
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_launch_gmm_fullcov(const float* Y, int64_t n, int d, int k, const float* theta, int c, float* logp,
                           float* grad, float* ws, hipStream_t s, const double* logp_ref, const void* packed) {
  if (d < 1 || d > DP || k < 1 || k > 16) return BVI_ERR_UNSUPPORTED;
  if (((uintptr_t)ws & 7) != 0 || ((uintptr_t)packed & 15) != 0) return BVI_ERR_ARG;
  double* psum = (double*)ws;
  float* pmin = ws + 2 * (size_t)kStatBlocks * DP;
  float* pmax = pmin + (size_t)kStatBlocks * DP;
  float* center = pmax + (size_t)kStatBlocks * DP;
  float* cscale = center + DP;
  float* Ainv = ws + kStatFloats;
  float* bvec = Ainv + (size_t)c * k * DP * DP;
  float* blo = bvec + (size_t)c * k * DP;
  float* cst = blo + (size_t)c * k * DP;
  float* at = cst + (size_t)c * k;
  float* apc0 = at + (size_t)c * k;
  apc0 += (4 - (((uintptr_t)apc0 >> 2) & 3)) & 3;                         // 16-B aligned
  f16x8* apc = reinterpret_cast<f16x8*>(apc0);
  float* partials = apc0 + (size_t)c * k * (3 * 2 * 64 * 4);
  const int64_t ntiles = (n + FT - 1) / FT;
  const int nblk = fullcov_blocks(n, c, grad != nullptr), reps = fullcov_replicas(k), cap = fullcov_list_cap(n, nblk);
  const size_t pstride = (size_t)c * ((size_t)k * (1 + DP + DP * DP) + 1);
  float* tstat = partials + (size_t)nblk * (1 + reps) * pstride;
  const int64_t ngroups = (ntiles + 63) / 64;
  float* gstat = tstat + (size_t)ntiles * (DP + 1);
  float* anorm = gstat + (size_t)ngroups * (DP + 1);
  float* mudist = anorm + (size_t)c * k;
  float* lnorm = mudist + (size_t)c * k;
  float* gb0 = lnorm + (size_t)c * k;
  gb0 += ((uintptr_t)gb0 & 7) ? 1 : 0;                 // float2 entries
  float2* gbounds = reinterpret_cast<float2*>(gb0);
  unsigned short* tmask = reinterpret_cast<unsigned short*>(gb0 + 2 * (size_t)c * ngroups * k);
  float* tl0 = gb0 + 2 * (size_t)c * ngroups * k + ((size_t)c * ntiles + 1) / 2 + 2;
  tl0 += ((uintptr_t)tl0 & 7) ? 1 : 0;                                    // (tile, mask) entries: 8-B aligned
  int2* tlist = reinterpret_cast<int2*>(tl0);
  int* toff = reinterpret_cast<int*>(tlist + (size_t)c * nblk * cap);
  int* any_single = toff + (size_t)c * nblk * kListOffs;
  any_single += (4 - (((uintptr_t)any_single >> 2) & 3)) & 3;            // 16-B aligned; the planes follow at + 16 B
  unsigned char* phi = reinterpret_cast<unsigned char*>(any_single + 4);
  unsigned char* plo = phi + (size_t)ntiles * FT * DP * 2;
  if (packed) {             // the rows' own numbers come from the packed buffer; nothing of them is computed here
    const PackedLayout L = packed_layout(n);
    unsigned char* pk = const_cast<unsigned char*>(static_cast<const unsigned char*>(packed));
    center = reinterpret_cast<float*>(pk + L.center);
    cscale = reinterpret_cast<float*>(pk + L.cscale);
    tstat = reinterpret_cast<float*>(pk + L.tstat);
    gstat = reinterpret_cast<float*>(pk + L.gstat);
    phi = pk + L.phi;
    plo = pk + L.plo;
  }
  // BVI_FULLCOV_F32=1 selects the exact-f32 kernel (v_mfma_f32_16x16x4_f32), kept as the reference the
  // f16-piece kernel is validated against
  static const bool use_f32 = getenv("BVI_FULLCOV_F32") && atoi(getenv("BVI_FULLCOV_F32")) != 0;
  const int nb = fullcov_blocks(n, c, grad != nullptr);
  int nb_total = nb;
  dim3 grid(nb, c);
  if (use_f32) {
    hipLaunchKernelGGL(fullcov_prepare, dim3(c * k), dim3(DP), 0, s, theta, c, k, d, (const float*)nullptr, Ainv,
                       bvec, (float*)nullptr, cst);
    BVI_HIP_CHECK_LAUNCH();
    const int kq = bvi_ceil_div(k, 4);
    const size_t lds = ((size_t)k * DP * LS + (size_t)k * DP + TR * LS + 4 * kq * TR + TR + 4 * kq) * sizeof(float);
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
  } else {
    static const bool no_prune = getenv("BVI_FULLCOV_NO_PRUNE") && atoi(getenv("BVI_FULLCOV_NO_PRUNE")) != 0;
    const bool prune = k > 1 && ntiles > 0 && !no_prune;       // one component: nothing to leave out
    if (!packed) {
      const int sblocks = (int)(ntiles / 4 < 1 ? 1 : (ntiles / 4 > kStatBlocks ? kStatBlocks : ntiles / 4));
      hipLaunchKernelGGL(fullcov_tilestats, dim3(sblocks), dim3(256), 0, s, Y, n, d, ntiles,
                         prune ? tstat : (float*)nullptr, psum, pmin, pmax);
      hipLaunchKernelGGL(fullcov_colstats_final, dim3(1), dim3(1024), 0, s, (const double*)psum, (const float*)pmin,
                         (const float*)pmax, sblocks, n, d, center, cscale);
    }
    hipLaunchKernelGGL(fullcov_prepare, dim3(c * k), dim3(DP), 0, s, theta, c, k, d, (const float*)center, Ainv,
                       bvec, blo, cst);
    BVI_HIP_CHECK_LAUNCH();
    if (prune) {
      hipLaunchKernelGGL(fullcov_anorm, dim3(c * k), dim3(256), 0, s, theta, k, d, (const float*)center, (const float*)Ainv,
                         anorm, mudist, lnorm);
      if (!packed)
        hipLaunchKernelGGL(fullcov_groupstats, dim3((unsigned)ngroups), dim3(64), 0, s, (const float*)tstat, ntiles, gstat);
      hipLaunchKernelGGL(fullcov_group_bounds, dim3((unsigned)bvi_ceil_div(ngroups, 64), c, k), dim3(64), 0, s,
                         (const float*)gstat, ngroups, k, (const float*)Ainv, (const float*)bvec, (const float*)cst,
                         (const float*)anorm, (const float*)center, gbounds);
      hipLaunchKernelGGL(fullcov_bounds, dim3((unsigned)bvi_ceil_div(ntiles, 64), c), dim3(64), 0, s, (const float*)tstat,
                         ntiles, k, d, theta, (const float*)Ainv, (const float*)bvec, (const float*)cst, (const float*)anorm,
                         (const float*)mudist, (const float*)lnorm, (const float*)center, c, (const float2*)gbounds, ngroups,
                         tmask);
      BVI_HIP_CHECK_LAUNCH();
    }
    const unsigned short* tmask_arg = prune ? tmask : nullptr;
    // one component: every tile is a single-component tile; no pruning otherwise: none is
    const bool lists_from_masks = prune || k == 1;
    if (!packed) {
      hipError_t me = hipMemsetAsync(any_single, 0, sizeof(int), s);
      if (me != hipSuccess) return (int)me;
    }
    hipLaunchKernelGGL(fullcov_tilelists, dim3(nblk, c), dim3(64), 0, s, lists_from_masks ? tmask_arg : (const unsigned short*)nullptr,
                       ntiles, lists_from_masks ? k : 16, cap, tlist, toff, any_single, grad != nullptr ? 1 : 0);
    hipLaunchKernelGGL(fullcov_apieces, dim3(c * k), dim3(64), 0, s, (const float*)Ainv, (const float*)cscale, apc, at);
    if (ntiles > 0 && !packed) {
      const int pblocks = (int)(ntiles > 4096 ? 4096 : ntiles);
      hipLaunchKernelGGL(fullcov_planes, dim3(pblocks), dim3(256), 0, s, Y, n, d, ntiles, (const float*)center,
                         (const float*)cscale, (const int*)any_single, phi, plo);
    }
    BVI_HIP_CHECK_LAUNCH();
#define BVI_FC16D(KQ, GRAD, PRUNE)                                                                                  \
  hipLaunchKernelGGL((fullcov_f16_kernel<KQ, GRAD, PRUNE>), grid, dim3(64 * FW), 0, s, Y, n, d, k, (const float*)Ainv,    \
                     (const float*)bvec, (const float*)blo, (const float*)cst, (const float*)center, (const float*)cscale, \
                     c, partials, tmask_arg, (const int2*)tlist, (const int*)toff, cap)
#define BVI_FC1(GRAD, FEW)                                                                                           \
  hipLaunchKernelGGL((fullcov_single_kernel<GRAD, FEW>), grid, dim3(64 * FW), 0, s, n, d, k, (const float*)bvec,          \
                     (const float*)blo, (const float*)cst, (const f16x8*)apc, (const float*)at, c,                          \
                     partials + (size_t)nblk * pstride, (const int2*)tlist, (const int*)toff, cap,                         \
                     (const unsigned char*)phi, (const unsigned char*)plo)
#define BVI_FC16(KQ, GRAD)                                                                                          \
  BVI_FC16D(KQ, GRAD, false);                                                                                        \
  if (prune) BVI_FC16D(KQ, GRAD, true);                                                                              \
  BVI_FC1(GRAD, false);                                                                                                \
  if (prune) BVI_FC1(GRAD, true)
    if (k <= FW) {
      if (grad) { BVI_FC16(1, true); } else { BVI_FC16(1, false); }
    } else {
      if (grad) { BVI_FC16(2, true); } else { BVI_FC16(2, false); }
    }
#undef BVI_FC16
#undef BVI_FC1
#undef BVI_FC16D
    nb_total = nblk * (1 + reps);
  }
  BVI_HIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(fullcov_finalize, dim3(c * k), dim3(8 * DP), 0, s, (const float*)partials, nb_total, n, d, k, theta,
                     (const float*)Ainv, use_f32 ? (const float*)nullptr : (const float*)cst, c, logp, grad, logp_ref,
                     use_f32 ? (const int*)nullptr : (const int*)toff, nb);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}
