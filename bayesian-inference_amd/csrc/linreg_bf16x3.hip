// Linear-regression raw sums (S2, S1, G) for many chains on the bf16 matrix
// cores with split operands: every f32 operand v is carried as bf16 pieces
// p0 = bf16(v), p1 = bf16(v - p0), p2 = bf16(v - p0 - p1) and a product as the
// sum of piece products, f32 accumulate:
//   2 pieces, 3 products (p0p0 + p1p0 + p0p1):            ~2^-18 relative per product
//   3 pieces, 6 products (+ p1p1 + p2p0 + p0p2):          ~2^-24, i.e. f32 grade
// The residual GEMM (X W^T, whose result cancels against y) takes NP pieces
// (template: 2 or 3); the gradient GEMM (R^T X) always takes 2.
// This is the chains x features x rows contraction of the likelihood node
// tfd.Normal(beta_0 + X w, scale) (bayes_vi/model/model.py:194-203,
// README.md:754-756) and of its gradient, fused so that the [chains, rows]
// residual matrix never leaves registers.
//
// Workgroup = 8 waves = 256 chains (wave w owns chains 32w..32w+31) x a strided
// set of 64-row tiles.  Per tile:
//   stage   X tile f32 (global, coalesced 32 B / thread) -> registers -> split ->
//           two row-major bf16 LDS images (hi, lo), double buffered, one barrier
//           per tile; y tile -> LDS.
//   GEMM1   R[32 rows, 32 chains] = (y - b0) + X (-W)^T   (v_mfma_f32_32x32x16_bf16,
//           A = X rows from LDS (ds_read_b128), B = -W fragments resident in VGPRs)
//   VALU    S1 += R, S2 += R^2, R -> bf16 hi/lo in the accumulator's own layout
//   GEMM2   G[32 chains, 32 feats] += R^T X: the accumulator registers 8s..8s+7
//           of a lane ARE the A fragment of k-step s (k = rows, permuted as
//           row = 16s + 8(j>>2) + 4(lane>>5) + (j&3)); B = X read column-wise from
//           the same LDS image with ds_read_b64_tr_b16 (4 rows x 16 cols per 16 lanes).
// Raw sums go to the workspace as [row_block][chain][S2, S1, G_0..G_{d-1}].
#include "bvi_common.h"

namespace {

constexpr int kThreads = 512;
constexpr int kChainsPerBlock = 256;
constexpr int kTileRows = 64;

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ void split8(const float* v, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}

template <int NP>
__device__ __forceinline__ void splitn(const float* v, bf16x8 (&p)[NP]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float rem = v[j];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const __bf16 h = (__bf16)rem;
      p[i][j] = h;
      rem -= (float)h;
    }
  }
}

template <int D, int NP>
__global__ __launch_bounds__(kThreads) void linreg_bf16x3_kernel(const float* __restrict__ X,
                                                                 const float* __restrict__ y, int64_t n,
                                                                 const float* __restrict__ theta, int C,
                                                                 int ncb, int nrb,
                                                                 float* __restrict__ partials) {
  constexpr int KS = D / 16;        // k-steps of GEMM1
  constexpr int NT = D / 32;        // 32-feature output tiles of GEMM2
  constexpr int SA = 2 * D + 16;    // LDS row stride in bytes (16-B pad: conflict-free ds_read_b128)
  constexpr int IMG = kTileRows * SA;
  constexpr int BUF = NP * IMG + kTileRows * 4;
  constexpr int TPR = D / 8;        // staging threads per row
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // blocks that share a row block sit on the same XCD (block ids are dealt
  // round-robin over the 8 XCDs) so the X tile is an L2 hit for all but one.
  const int x8 = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int cbi = jj % ncb, rbi = (jj / ncb) * 8 + x8;
  const int chain = cbi * kChainsPerBlock + 32 * wave + r;
  const bool cvalid = chain < C;

  // -W fragments (B operand of GEMM1) and b0, resident for the whole kernel
  bf16x8 wp[KS][NP];
  float b0 = 0.f;
  {
    const float* th = theta + (int64_t)(cvalid ? chain : 0) * (D + 2);
    if (cvalid) b0 = th[0];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = cvalid ? -th[1 + 16 * ks + 8 * h + j] : 0.f;
      splitn<NP>(v, wp[ks]);
    }
  }

  f32x16 G[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int i = 0; i < 16; ++i) G[nt][i] = 0.f;
  float s1 = 0.f, s2 = 0.f;

  // staging assignment: thread -> (row, 8 consecutive features)
  const int srow = tid / TPR, scol = (tid % TPR) * 8;
  const bool sact = srow < kTileRows;
  const int64_t ntiles = (n + kTileRows - 1) / kTileRows;

  float xr[8], yr = 0.f;
  auto gload = [&](int64_t tile) {
    const int64_t row = tile * kTileRows + srow;
    if (sact && row < n) {
      const f32x4* p = reinterpret_cast<const f32x4*>(X + row * D + scol);
      const f32x4 a = p[0], b = p[1];
#pragma unroll
      for (int j = 0; j < 4; ++j) { xr[j] = a[j]; xr[4 + j] = b[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) xr[j] = 0.f;
    }
    if (tid < kTileRows) {
      const int64_t yrow = tile * kTileRows + tid;
      yr = yrow < n ? y[yrow] : 0.f;
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* base = lds + buf * BUF;
    if (sact) {
      bf16x8 xp[NP];
      splitn<NP>(xr, xp);
#pragma unroll
      for (int i = 0; i < NP; ++i) *reinterpret_cast<bf16x8*>(base + i * IMG + srow * SA + scol * 2) = xp[i];
    }
    if (tid < kTileRows) reinterpret_cast<float*>(base + NP * IMG)[tid] = yr;
  };

  int64_t tile = rbi;
  if (tile < ntiles) {
    gload(tile);
    lstore(0);
  }
  __syncthreads();
  int buf = 0;
  for (; tile < ntiles; tile += nrb) {
    const int64_t tnext = tile + nrb;
    if (tnext < ntiles) gload(tnext);

    const unsigned char* base = lds + buf * BUF;
    const float* yb = reinterpret_cast<const float*>(base + NP * IMG);
    const bool full = tile * kTileRows + kTileRows <= n;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      // ---- GEMM1: residuals of 32 rows x 32 chains ----
      f32x16 z;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(yb + 32 * st + 8 * gq + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) z[4 * gq + e] = yv[e] - b0;
      }
      if (!full) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = 32 * st + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (tile * kTileRows + row >= n) z[i] = 0.f;
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int off = (32 * st + r) * SA + (16 * ks + 8 * h) * 2;
        bf16x8 a[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) a[i] = *reinterpret_cast<const bf16x8*>(base + i * IMG + off);
        // piece products of total order <= NP-1 (piece i carries ~2^-8i of the value)
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
          for (int j = 0; j + i < NP; ++j)
            z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], wp[ks][j], z, 0, 0, 0);
      }
      // ---- S1, S2 and the bf16 split of R ----
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        a1 += z[i];
        a2 = fmaf(z[i], z[i], a2);
      }
      s1 += a1;
      s2 += a2;
      bf16x8 rh[2], rl[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = z[8 * s + j];
        split8(v, rh[s], rl[s]);
      }
      // ---- GEMM2: G += R^T X ----
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int row0 = 32 * st + 16 * s + 4 * h + ((lane & 15) >> 2);
          const int col = 32 * nt + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
          const int off0 = row0 * SA + col * 2;
          const int off1 = off0 + 8 * SA;
          const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off0));
          const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off1));
          const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + IMG + off0));
          const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + IMG + off1));
          const bf16x8 xh = __builtin_bit_cast(bf16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
          const bf16x8 xl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
          G[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rh[s], xh, G[nt], 0, 0, 0);
          G[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rl[s], xh, G[nt], 0, 0, 0);
          G[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rh[s], xl, G[nt], 0, 0, 0);
        }
      }
    }

    if (tnext < ntiles) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- write raw sums of this (row block, chain block) ----
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  if (cvalid && h == 0) {
    float* p = partials + ((int64_t)rbi * C + chain) * (D + 2);
    p[0] = s2;
    p[1] = s1;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = cbi * kChainsPerBlock + 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (c < C) partials[((int64_t)rbi * C + c) * (D + 2) + 2 + 32 * nt + r] = G[nt][i];
    }
}

}  // namespace

static void bf16x3_grid(int c, int* ncb, int* nrb) {
  *ncb = bvi_ceil_div(c, kChainsPerBlock);
  int per8 = 32 / *ncb;  // 8 * ncb * per8 ~ 256 workgroups: one per CU
  if (per8 < 1) per8 = 1;
  *nrb = 8 * per8;
}

int bvi_linreg_bf16x3_blocks(int64_t n, int c) {
  int ncb, nrb;
  bf16x3_grid(c, &ncb, &nrb);
  (void)n;
  return nrb;
}

int bvi_launch_linreg_bf16x3(const float* X, const float* y, int64_t n, int d, const float* theta, int c,
                             int pieces, float* ws, int* nb_out, hipStream_t s) {
  if (((uintptr_t)X & 15) != 0) return BVI_ERR_ARG;
  int ncb, nrb;
  bf16x3_grid(c, &ncb, &nrb);
  *nb_out = nrb;
  dim3 grid(ncb * nrb);
  if (d == 64 && pieces == 2)
    hipLaunchKernelGGL((linreg_bf16x3_kernel<64, 2>), grid, dim3(kThreads), 0, s, X, y, n, theta, c, ncb, nrb, ws);
  else if (d == 64 && pieces == 3)
    hipLaunchKernelGGL((linreg_bf16x3_kernel<64, 3>), grid, dim3(kThreads), 0, s, X, y, n, theta, c, ncb, nrb, ws);
  else if (d == 32 && pieces == 2)
    hipLaunchKernelGGL((linreg_bf16x3_kernel<32, 2>), grid, dim3(kThreads), 0, s, X, y, n, theta, c, ncb, nrb, ws);
  else if (d == 32 && pieces == 3)
    hipLaunchKernelGGL((linreg_bf16x3_kernel<32, 3>), grid, dim3(kThreads), 0, s, X, y, n, theta, c, ncb, nrb, ws);
  else
    return BVI_ERR_UNSUPPORTED;
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}
