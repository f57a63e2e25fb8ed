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

// PACKED: X and y come pre-split (bvi_linreg_pack): bf16 piece planes [3][n_pad][D] followed by
// the row extensions [n_pad][8], n_pad = n rounded up to the tile with zero rows, so staging
// is a plain 16-B copy with no arithmetic and no tail handling.  Otherwise X, y are f32 and
// are split in registers on the way to LDS.
template <int D, int NP, bool PACKED>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void linreg_bf16x3_kernel(
    const float* __restrict__ X, const float* __restrict__ y, const unsigned char* __restrict__ packed, int64_t n,
    const float* __restrict__ theta, int C, int ncb, int nrb, float* __restrict__ partials) {
  constexpr int KS = D / 16;        // k-steps of GEMM1
  constexpr int NT = D / 32;        // 32-feature output tiles of GEMM2
  constexpr int SA = 2 * D + 16;    // LDS row stride in bytes (16-B pad: conflict-free ds_read_b128)
  constexpr int IMG = kTileRows * SA;
  constexpr int EXT = kTileRows * 16;        // per row [y0 y1 y2 1 1 1 0 0] bf16
  constexpr int BUF = NP * IMG + EXT;
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

  // -W fragments (B operand of GEMM1), resident for the whole kernel, and the
  // "extension" fragment [1 1 1 -b0_0 -b0_1 -b0_2 0 0] that turns the k-step over
  // the LDS row extension [y_0 y_1 y_2 1 1 1 0 0] into y - b0 (3-piece, f32 grade).
  bf16x8 wp[KS][NP], wext;
  {
    const float* th = theta + (int64_t)(cvalid ? chain : 0) * (D + 2);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (cvalid ? -1.f : 0.f) * th[1 + 16 * ks + 8 * h + j];
      splitn<NP>(v, wp[ks]);
    }
    float rem = (cvalid ? -1.f : 0.f) * th[0];
#pragma unroll
    for (int j = 0; j < 8; ++j) wext[j] = (__bf16)0.f;
    if (h == 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        wext[j] = (__bf16)1.f;
        const __bf16 p = (__bf16)rem;
        wext[3 + j] = p;
        rem -= (float)p;
      }
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
  // with D = 64 every thread stages (no exec-masked block for the compiler to sink loads into)
  const bool sact = (kThreads / TPR == kTileRows) ? true : (srow < kTileRows);
  const int64_t ntiles = (n + kTileRows - 1) / kTileRows;

  f32x4 xa = {0.f, 0.f, 0.f, 0.f}, xb = {0.f, 0.f, 0.f, 0.f};
  float yr = 0.f;
  bool xok = false, yok = false;
  bf16x8 xp[NP], ye;
  const int64_t n_pad = ntiles * kTileRows;
  // Prefetch of one tile into registers.  Branch-free: out-of-range threads read row 0
  // (always valid here); the loaded values are first touched in stage_split / lstore so
  // the wait lands there, not here.
  auto gload = [&](int64_t tile) {
    if constexpr (PACKED) {
      const int64_t row = (tile < ntiles ? tile : 0) * kTileRows + (sact ? srow : 0);
#pragma unroll
      for (int i = 0; i < NP; ++i)
        xp[i] = *reinterpret_cast<const bf16x8*>(packed + (((int64_t)i * n_pad + row) * D + scol) * 2);
      ye = *reinterpret_cast<const bf16x8*>(packed + (int64_t)3 * n_pad * D * 2 + row * 16);
    } else {
      const int64_t row = tile * kTileRows + srow;
      xok = sact && row < n;
      const f32x4* p = reinterpret_cast<const f32x4*>(X + (xok ? row : 0) * D + (sact ? scol : 0));
      xa = p[0];
      xb = p[1];
      yok = xok;   // every staging thread of a row carries its y
      yr = y[yok ? row : 0];
    }
  };
  // split the staged registers into bf16 pieces (pure VALU: overlaps the MFMAs) ...
  auto stage_split = [&]() {
    if constexpr (!PACKED) {
      float xr[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xr[j] = xok ? xa[j] : 0.f;
        xr[4 + j] = xok ? xb[j] : 0.f;
      }
      splitn<NP>(xr, xp);
      // rows past the end get an all-zero extension: their residual is exactly 0
      float rem = yok ? yr : 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const __bf16 p = (__bf16)rem;
        ye[j] = p;
        rem -= (float)p;
        ye[3 + j] = (__bf16)(yok ? 1.f : 0.f);
      }
      ye[6] = (__bf16)0.f;
      ye[7] = (__bf16)0.f;
    }
  };
  // ... and write them to the LDS images of buffer `buf`.  All staging threads of a row
  // write its 16-B extension (same bytes from several threads: benign, and branch-free).
  auto lstore = [&](int buf) {
    unsigned char* base = lds + buf * BUF;
    if (sact) {
#pragma unroll
      for (int i = 0; i < NP; ++i) *reinterpret_cast<bf16x8*>(base + i * IMG + srow * SA + scol * 2) = xp[i];
      *reinterpret_cast<bf16x8*>(base + NP * IMG + srow * 16) = ye;
    }
  };

  // GEMM1 of one 32-row sub-tile: residuals R[32 rows, 32 chains] in the accumulator
  auto gemm1 = [&](const unsigned char* base, int st) -> f32x16 {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    // both lane halves read the same 16 B: the B fragment is zero for k >= 8
    const bf16x8 ae = *reinterpret_cast<const bf16x8*>(base + NP * IMG + (32 * st + r) * 16);
    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ae, wext, z, 0, 0, 0);
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
    return z;
  };
  // S1, S2 and the 2-piece bf16 split of R, in the accumulator's own layout
  auto post = [&](const f32x16& z, bf16x8 (&rh)[2], bf16x8 (&rl)[2]) {
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      a1 += z[i];
      a2 = fmaf(z[i], z[i], a2);
    }
    s1 += a1;
    s2 += a2;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = z[8 * s + j];
      split8(v, rh[s], rl[s]);
    }
  };
  // GEMM2 of one sub-tile: G += R^T X
  auto gemm2 = [&](const unsigned char* base, int st, const bf16x8 (&rh)[2], const bf16x8 (&rl)[2]) {
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
  };

  int64_t tile = rbi;
  if (tile < ntiles) {
    gload(tile);
    stage_split();
    lstore(0);
  }
  __syncthreads();
  int buf = 0;
  for (; tile < ntiles; tile += nrb) {
    const int64_t tnext = tile + nrb;
    gload(tnext);   // past the end: loads nothing, stages zeros
    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch at the top of the tile

    const unsigned char* base = lds + buf * BUF;
    // two independent residual chains; the VALU work on the first overlaps the
    // MFMAs of the second, the VALU work on the second those of the first GEMM2
    const f32x16 z0 = gemm1(base, 0);
    const f32x16 z1 = gemm1(base, 1);
    bf16x8 rh0[2], rl0[2], rh1[2], rl1[2];
    post(z0, rh0, rl0);
    gemm2(base, 0, rh0, rl0);
    // the prefetched tile has landed by now; the other buffer is free since the
    // last barrier (past the end this stores zeros that nobody reads)
    stage_split();
    lstore(buf ^ 1);
    post(z1, rh1, rl1);
    gemm2(base, 1, rh1, rl1);
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

// One thread per (row, 8 features): f32 -> three bf16 piece planes; the thread of the
// first feature group also writes the row extension.  Rows n..n_pad-1 are written as zeros.
template <int D>
__global__ void linreg_pack_kernel(const float* __restrict__ X, const float* __restrict__ y, int64_t n,
                                   int64_t n_pad, unsigned char* __restrict__ packed) {
  constexpr int TPR = D / 8;
  const int64_t total = n_pad * TPR;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / TPR;
    const int scol = (int)(idx % TPR) * 8;
    const bool ok = row < n;
    float v[8];
    if (ok) {
      const f32x4* p = reinterpret_cast<const f32x4*>(X + row * D + scol);
      const f32x4 a = p[0], b = p[1];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    bf16x8 xp[3];
    splitn<3>(v, xp);
#pragma unroll
    for (int i = 0; i < 3; ++i)
      *reinterpret_cast<bf16x8*>(packed + (((int64_t)i * n_pad + row) * D + scol) * 2) = xp[i];
    if (scol == 0) {
      bf16x8 ye;
      float rem = ok ? y[row] : 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const __bf16 pz = (__bf16)rem;
        ye[j] = pz;
        rem -= (float)pz;
        ye[3 + j] = (__bf16)(ok ? 1.f : 0.f);
      }
      ye[6] = (__bf16)0.f;
      ye[7] = (__bf16)0.f;
      *reinterpret_cast<bf16x8*>(packed + (int64_t)3 * n_pad * D * 2 + row * 16) = ye;
    }
  }
}

}  // namespace

static int64_t pad_rows(int64_t n) { return (n + kTileRows - 1) / kTileRows * kTileRows; }

size_t bvi_linreg_packed_size(int64_t n, int d) {
  const int64_t np = pad_rows(n);
  return (size_t)(np * d * 6 + np * 16);
}

int bvi_launch_linreg_pack(const float* X, const float* y, int64_t n, int d, void* packed, hipStream_t s) {
  if (((uintptr_t)X & 15) != 0 || ((uintptr_t)packed & 15) != 0) return BVI_ERR_ARG;
  const int64_t np = pad_rows(n);
  if (np == 0) return BVI_OK;
  const int64_t total = np * (d / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (d == 64)
    hipLaunchKernelGGL(linreg_pack_kernel<64>, dim3(blocks), dim3(256), 0, s, X, y, n, np, (unsigned char*)packed);
  else if (d == 32)
    hipLaunchKernelGGL(linreg_pack_kernel<32>, dim3(blocks), dim3(256), 0, s, X, y, n, np, (unsigned char*)packed);
  else
    return BVI_ERR_UNSUPPORTED;
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

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

// X, y (f32, split in the kernel) or `packed` (from bvi_launch_linreg_pack); exactly one is given.
int bvi_launch_linreg_bf16x3(const float* X, const float* y, const void* packed, int64_t n, int d,
                             const float* theta, int c, int pieces, float* ws, int* nb_out, hipStream_t s) {
  if (!packed && ((uintptr_t)X & 15) != 0) return BVI_ERR_ARG;
  if (packed && ((uintptr_t)packed & 15) != 0) return BVI_ERR_ARG;
  int ncb, nrb;
  bf16x3_grid(c, &ncb, &nrb);
  *nb_out = nrb;
  dim3 grid(ncb * nrb);
  const unsigned char* pk = (const unsigned char*)packed;
#define BVI_BF16_CASE(DD, NPP)                                                                              \
  if (d == DD && pieces == NPP) {                                                                           \
    if (pk)                                                                                                 \
      hipLaunchKernelGGL((linreg_bf16x3_kernel<DD, NPP, true>), grid, dim3(kThreads), 0, s, X, y, pk, n,    \
                         theta, c, ncb, nrb, ws);                                                           \
    else                                                                                                    \
      hipLaunchKernelGGL((linreg_bf16x3_kernel<DD, NPP, false>), grid, dim3(kThreads), 0, s, X, y, pk, n,   \
                         theta, c, ncb, nrb, ws);                                                           \
    BVI_HIP_CHECK_LAUNCH();                                                                                 \
    return BVI_OK;                                                                                          \
  }
  BVI_BF16_CASE(64, 3)
  BVI_BF16_CASE(64, 2)
  BVI_BF16_CASE(32, 3)
  BVI_BF16_CASE(32, 2)
#undef BVI_BF16_CASE
  return BVI_ERR_UNSUPPORTED;
}
