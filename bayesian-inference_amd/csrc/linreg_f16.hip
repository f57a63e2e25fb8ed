// Linear-regression raw sums for many chains on the f16 matrix cores, operands split into f16
// pieces of range-normalised values (11 significant bits each, plus one from round-to-nearest).
// Residual GEMM, 4 products: X = xh + xl (error <= 1 f32 ulp, independent from row to row, so it
// averages out over rows) times W = w1 + w2 + w3 (exact): xh.w1, xl.w1, xh.w2, xh.w3.  The third
// piece of W matters although it is 2^-22 of W: an error in W is the same for every row, so it
// acts like evaluating the gradient at a shifted parameter and grows with n, not sqrt(n).
// Gradient GEMM, 3 products: R = rh + rl times X: rh.xh, rl.xh, rh.xl.
// 29 MFMAs per 32 rows x 32 chains at D = 64 where the all-bf16 kernel (linreg_bf16x3.hip, 6 + 3
// products) needs 37, on half the bytes (two 2-byte planes of X = the size of the f32 matrix).
// Same likelihood node as there: tfd.Normal(beta_0 + X w, scale)
// (bayes_vi/model/model.py:194-203, README.md:754-756).
//
// f16 has 11 significant bits but only 5 exponent bits, so every f16 operand is brought to the top
// of its range by an exact power-of-two scale:
//   x'_id = x_id * s_d,    s_d = 2^(13 - floor(log2 max_i |x_id|))            (per column, at pack time;
//                                                                          0 for an all-zero column)
//   w''_cd = -w_cd / s_d * t_c,  t_c = 2^(13 - floor(log2 max_d |w_cd / s_d|))  (per chain, per call)
// so that |x'|, |w''| < 2^14 and z' = sum_d x'_id w''_cd = -t_c x_i.w_c accumulates in f32; the low
// pieces then sit 2^-11 below, far above the f16 subnormal threshold for every element that
// matters to the sum.  r = (y - b0) + z' / t_c, with y - b0 from one bf16 k-step over the row
// extension (as in linreg_bf16x3.hip).  The residuals enter the gradient GEMM as
//   r'_ic = r_ic * u_c,    u_c = 2^(14 - floor(log2 (max|y| + |b0| + D 2^14 max_d |w_cd / s_d|)))
// : the bound holds for every row, so |r'| < 2^15, and it is loose by a few binades only, which
// leaves ~20 binades of f16 range below a typical residual.  G'_cd = sum_i r'_ic x'_id is rescaled
// by 1 / (u_c s_d) in the finalize kernel (exact: powers of two).
// S1 = sum_i r_i is not accumulated at all: it equals sum y - n b0 - w . (column sums of X), with
// the data sums taken once, in double, at pack time.
//
// Packed dataset (bvi_linreg_pack, format BVI_PACK_F16), 16-B aligned throughout:
//   header [1024 B]: float colscale[64]; double colsum[64]; double ysum; float ymax
//   planes [2][n_pad][D] f16: hi(x'), lo(x')
//   ext    [n_pad][8] bf16: [y0 y1 y2 1 1 1 0 0]   (all zero for padding rows)
#include "bvi_common.h"

namespace {

constexpr int kThreads = 512;
constexpr int kChainsPerBlock = 256;
constexpr int kTileRows = 64;
constexpr int kHeader = 1024;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct Header {
  float colscale[64];
  double colsum[64];
  double ysum;
  float ymax;
};
static_assert(sizeof(Header) <= kHeader, "header must fit");

// scale of the residuals for the gradient GEMM: |r| <= ymax + |b0| + D * 2^14 * max|w / colscale|
__device__ __forceinline__ float resid_scale(float amax_v, float b0, float ymax, int d) {
  const float bound = ymax + fabsf(b0) + (float)d * 16384.f * amax_v;
  if (!(bound > 0.f) || !isfinite(bound)) return 1.f;
  return ldexpf(1.f, min(max(14 - ilogbf(bound), -126), 126));
}

// w / s_d; an all-zero column (s_d = 0) takes no part in the chain's range (0, or NaN for a
// non-finite weight, as 0 * w would give)
__device__ __forceinline__ float scaled_weight(float w, float colscale) {
  return colscale > 0.f ? w / colscale : w * 0.f;
}

__device__ __forceinline__ float pow2_scale(float amax) {
  // 2^(13 - floor(log2 amax)), kept inside the normal f32 range; 1 for zero / non-finite input
  if (!(amax > 0.f) || !isfinite(amax)) return 1.f;
  return ldexpf(1.f, min(max(13 - ilogbf(amax), -126), 126));
}

template <int D, int NP1>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void linreg_f16_kernel(
    const unsigned char* __restrict__ packed, int64_t n, const float* __restrict__ theta, int C, int ncb, int nrb,
    float* __restrict__ partials) {
  constexpr int KS = D / 16;
  constexpr int NT = D / 32;
  constexpr int SA = 2 * D + 16;
  constexpr int IMG = kTileRows * SA;
  constexpr int EXT = kTileRows * 16;
  constexpr int BUF = 2 * IMG + EXT;
  constexpr int TPR = D / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int x8 = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int cbi = jj % ncb, rbi = (jj / ncb) * 8 + x8;
  const int chain = cbi * kChainsPerBlock + 32 * wave + r;
  const bool cvalid = chain < C;
  const Header* hdr = reinterpret_cast<const Header*>(packed);
  const int64_t ntiles = (n + kTileRows - 1) / kTileRows;
  const int64_t n_pad = ntiles * kTileRows;
  const unsigned char* planes = packed + kHeader;
  const unsigned char* ext = planes + (int64_t)2 * n_pad * D * 2;

  // scaled -W fragments in f16 pieces, the chain's 1/t and residual scale u, and the bf16 extension
  // fragment for y - b0
  f16x8 wh[KS], wl[KS], w3[KS];
  bf16x8 wext;
  float inv_t, u;
  {
    const float* th = theta + (int64_t)(cvalid ? chain : 0) * (D + 2);
    float v[KS][8], amax = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int f = 16 * ks + 8 * h + j;
        v[ks][j] = (cvalid ? -1.f : 0.f) * scaled_weight(th[1 + f], hdr->colscale[f]);
        amax = fmaxf(amax, fabsf(v[ks][j]));
      }
    amax = fmaxf(amax, __shfl_xor(amax, 32, 64));     // the two lane halves hold the chain's 64 features
    const float t = pow2_scale(amax);
    inv_t = 1.f / t;
    u = resid_scale(amax, cvalid ? th[0] : 0.f, hdr->ymax, D);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float q = v[ks][j] * t;
        const _Float16 hi = (_Float16)q;
        wh[ks][j] = hi;
        const _Float16 lo = (_Float16)(q - (float)hi);
        wl[ks][j] = lo;
        w3[ks][j] = (_Float16)((q - (float)hi) - (float)lo);
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
  float s2 = 0.f;

  const int srow = tid / TPR, scol = (tid % TPR) * 8;
  const bool sact = (kThreads / TPR == kTileRows) ? true : (srow < kTileRows);

  s16x8 xp[2], ye;
  auto gload = [&](int64_t tile) {
    const int64_t row = (tile < ntiles ? tile : 0) * kTileRows + (sact ? srow : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      xp[i] = *reinterpret_cast<const s16x8*>(planes + (((int64_t)i * n_pad + row) * D + scol) * 2);
    ye = *reinterpret_cast<const s16x8*>(ext + row * 16);
  };
  auto lstore = [&](int buf) {
    unsigned char* base = lds + buf * BUF;
    if (sact) {
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<s16x8*>(base + i * IMG + srow * SA + scol * 2) = xp[i];
      *reinterpret_cast<s16x8*>(base + 2 * IMG + srow * 16) = ye;
    }
  };
  // residuals of one 32-row sub-tile: r = (y - b0) + z' / t
  auto gemm1 = [&](const unsigned char* base, int st) -> f32x16 {
    f32x16 z, yb;
#pragma unroll
    for (int i = 0; i < 16; ++i) { z[i] = 0.f; yb[i] = 0.f; }
    const bf16x8 ae = *reinterpret_cast<const bf16x8*>(base + 2 * IMG + (32 * st + r) * 16);
    yb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ae, wext, yb, 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int off = (32 * st + r) * SA + (16 * ks + 8 * h) * 2;
      const f16x8 ah = *reinterpret_cast<const f16x8*>(base + off);
      const f16x8 al = *reinterpret_cast<const f16x8*>(base + IMG + off);
      z = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[ks], z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[ks], z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[ks], z, 0, 0, 0);
      if (NP1 == 4) z = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w3[ks], z, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = fmaf(z[i], inv_t, yb[i]);
    return z;
  };
  auto post = [&](const f32x16& z, f16x8 (&rh)[2], f16x8 (&rl)[2]) {
    float a2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a2 = fmaf(z[i], z[i], a2);
    s2 += a2;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float q = z[8 * s + j] * u;
        const _Float16 hi = (_Float16)q;
        rh[s][j] = hi;
        rl[s][j] = (_Float16)(q - (float)hi);
      }
  };
  auto gemm2 = [&](const unsigned char* base, int st, const f16x8 (&rh)[2], const f16x8 (&rl)[2]) {
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
        const f16x8 xh = __builtin_bit_cast(f16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
        const f16x8 xl = __builtin_bit_cast(f16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
        G[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh[s], xh, G[nt], 0, 0, 0);
        G[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rl[s], xh, G[nt], 0, 0, 0);
        G[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh[s], xl, G[nt], 0, 0, 0);
      }
    }
  };

  int64_t tile = rbi;
  if (tile < ntiles) {
    gload(tile);
    lstore(0);
  }
  __syncthreads();
  int buf = 0;
  for (; tile < ntiles; tile += nrb) {
    gload(tile + nrb);   // past the end: re-reads tile 0, never used
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* base = lds + buf * BUF;
    const f32x16 z0 = gemm1(base, 0);
    const f32x16 z1 = gemm1(base, 1);
    f16x8 rh0[2], rl0[2], rh1[2], rl1[2];
    post(z0, rh0, rl0);
    gemm2(base, 0, rh0, rl0);
    lstore(buf ^ 1);
    post(z1, rh1, rl1);
    gemm2(base, 1, rh1, rl1);
    __syncthreads();
    buf ^= 1;
  }

  s2 += __shfl_xor(s2, 32, 64);
  if (cvalid && h == 0) {
    float* p = partials + ((int64_t)rbi * C + chain) * (D + 2);
    p[0] = s2;
    p[1] = 0.f;   // S1 comes from the packed data sums (finalize)
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = cbi * kChainsPerBlock + 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (c < C) partials[((int64_t)rbi * C + c) * (D + 2) + 2 + 32 * nt + r] = G[nt][i];
    }
}

// ---- pack: column statistics (two-stage, fixed order) then the split ----
// stage 1: block b reduces rows b, b + nblk, ... : per column abs-max and sum (double), sum and abs-max of y
template <int D>
__global__ void f16_colstats_kernel(const float* __restrict__ X, const float* __restrict__ y, int64_t n,
                                      float* __restrict__ pmax, double* __restrict__ psum) {
  // 256 threads = 256 / D row lanes x D columns
  constexpr int RL = 256 / D;
  const int col = threadIdx.x % D, rl = threadIdx.x / D;
  float amax = 0.f;
  float ymax = 0.f;
  double sum = 0.0, ysum = 0.0;
  for (int64_t row = (int64_t)blockIdx.x * RL + rl; row < n; row += (int64_t)gridDim.x * RL) {
    const float v = X[row * D + col];
    amax = fmaxf(amax, fabsf(v));
    sum += (double)v;
    if (col == 0) {
      ysum += (double)y[row];
      ymax = fmaxf(ymax, fabsf(y[row]));
    }
  }
  __shared__ float smax[256], symax[256];
  __shared__ double ssum[256], sy[256];
  smax[threadIdx.x] = amax;
  ssum[threadIdx.x] = sum;
  sy[threadIdx.x] = ysum;
  symax[threadIdx.x] = ymax;
  __syncthreads();
  if (rl == 0) {
    for (int q = 1; q < RL; ++q) {
      amax = fmaxf(amax, smax[q * D + col]);
      sum += ssum[q * D + col];
      if (col == 0) {
        ysum += sy[q * D];
        ymax = fmaxf(ymax, symax[q * D]);
      }
    }
    pmax[(int64_t)blockIdx.x * (D + 1) + col] = amax;
    if (col == 0) pmax[(int64_t)blockIdx.x * (D + 1) + D] = ymax;
    psum[(int64_t)blockIdx.x * (D + 1) + col] = sum;
    if (col == 0) psum[(int64_t)blockIdx.x * (D + 1) + D] = ysum;
  }
}

template <int D>
__global__ void f16_header_kernel(const float* __restrict__ pmax, const double* __restrict__ psum, int nblk,
                                    Header* __restrict__ hdr) {
  const int col = threadIdx.x;
  if (col < D) {
    float amax = 0.f;
    double sum = 0.0;
    for (int b = 0; b < nblk; ++b) {
      amax = fmaxf(amax, pmax[(int64_t)b * (D + 1) + col]);
      sum += psum[(int64_t)b * (D + 1) + col];
    }
    hdr->colscale[col] = amax > 0.f ? pow2_scale(amax) : 0.f;   // 0 marks an all-zero column
    hdr->colsum[col] = sum;
  } else if (col < 64) {
    hdr->colscale[col] = 0.f;
    hdr->colsum[col] = 0.0;
  }
  if (col == 0) {
    double ys = 0.0;
    float ym = 0.f;
    for (int b = 0; b < nblk; ++b) {
      ys += psum[(int64_t)b * (D + 1) + D];
      ym = fmaxf(ym, pmax[(int64_t)b * (D + 1) + D]);
    }
    hdr->ysum = ys;
    hdr->ymax = ym;
  }
}

template <int D>
__global__ void f16_pack_kernel(const float* __restrict__ X, const float* __restrict__ y, int64_t n,
                                  int64_t n_pad, unsigned char* __restrict__ packed) {
  constexpr int TPR = D / 8;
  const Header* hdr = reinterpret_cast<const Header*>(packed);
  unsigned char* planes = packed + kHeader;
  unsigned char* ext = planes + (int64_t)2 * n_pad * D * 2;
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
    f16x8 fh, fl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float q = v[j] * hdr->colscale[scol + j];
      const _Float16 hi = (_Float16)q;
      fh[j] = hi;
      fl[j] = (_Float16)(q - (float)hi);
    }
    const int64_t off = (row * D + scol) * 2;
    const int64_t plane = n_pad * D * 2;
    *reinterpret_cast<f16x8*>(planes + off) = fh;
    *reinterpret_cast<f16x8*>(planes + plane + off) = fl;
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
      *reinterpret_cast<bf16x8*>(ext + row * 16) = ye;
    }
  }
}

// finalize: fixed-order reduction over row blocks (double) + scaling; S1 from the packed data sums
__global__ void linreg_f16_finalize(const float* __restrict__ partials, int nb, int64_t n, int d,
                                      const unsigned char* __restrict__ packed, const float* __restrict__ theta,
                                      int C, float* __restrict__ logp, float* __restrict__ grad) {
  const int q2 = d + 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= C * q2) return;
  const int c = idx / q2, q = idx - c * q2;
  const Header* hdr = reinterpret_cast<const Header*>(packed);
  const float* th = theta + (int64_t)c * q2;
  double s = 0.0;
  if (q == 1) {
    s = hdr->ysum - (double)n * (double)th[0];
    for (int j = 0; j < d; ++j) s -= (double)th[1 + j] * hdr->colsum[j];
  } else {
    for (int b = 0; b < nb; ++b) s += (double)partials[((int64_t)b * C + c) * q2 + q];
  }
  const double sc = (double)th[d + 1];
  const double inv2 = 1.0 / (sc * sc), nn = (double)n;
  if (q == 0) {
    logp[c] = (float)(-0.5 * s * inv2 - nn * log(sc) - 0.5 * nn * BVI_LN2PI_D);
    if (grad) grad[(int64_t)c * q2 + d + 1] = (float)(-nn / sc + s * inv2 / sc);
  } else if (grad) {
    if (q >= 2) {   // undo the scales of the gradient GEMM: G' = u_c * s_d * G
      float amax = 0.f;
      for (int j = 0; j < d; ++j) amax = fmaxf(amax, fabsf(scaled_weight(th[1 + j], hdr->colscale[j])));
      const double cs = (double)hdr->colscale[q - 2];
      s = cs > 0.0 ? s / ((double)resid_scale(amax, th[0], hdr->ymax, d) * cs) : s * 0.0;
    }
    grad[(int64_t)c * q2 + (q - 1)] = (float)(s * inv2);
  }
}

int64_t pad_rows(int64_t n) { return (n + kTileRows - 1) / kTileRows * kTileRows; }
constexpr int kStatBlocks = 1024;

}  // namespace

size_t bvi_linreg_f16_packed_size(int64_t n, int d) {
  const int64_t np = pad_rows(n);
  return (size_t)(kHeader + np * d * 4 + np * 16);
}

// scratch for the column statistics of the pack step (bytes)
size_t bvi_linreg_f16_pack_scratch(int d) {
  return (size_t)kStatBlocks * (d + 1) * sizeof(float) + (size_t)kStatBlocks * (d + 1) * sizeof(double) + 16;
}

int bvi_launch_linreg_f16_pack(const float* X, const float* y, int64_t n, int d, void* packed, void* scratch,
                                 hipStream_t s) {
  if (((uintptr_t)X & 15) != 0 || ((uintptr_t)packed & 15) != 0 || ((uintptr_t)scratch & 7) != 0)
    return BVI_ERR_ARG;
  if (d != 32 && d != 64) return BVI_ERR_UNSUPPORTED;
  const int64_t np = pad_rows(n);
  double* psum = (double*)scratch;
  float* pmax = (float*)(psum + (size_t)kStatBlocks * (d + 1));
  const int64_t total = np * (d / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  if (d == 64) {
    hipLaunchKernelGGL(f16_colstats_kernel<64>, dim3(kStatBlocks), dim3(256), 0, s, X, y, n, pmax, psum);
    hipLaunchKernelGGL(f16_header_kernel<64>, dim3(1), dim3(64), 0, s, (const float*)pmax, (const double*)psum,
                       kStatBlocks, (Header*)packed);
    hipLaunchKernelGGL(f16_pack_kernel<64>, dim3(blocks), dim3(256), 0, s, X, y, n, np, (unsigned char*)packed);
  } else {
    hipLaunchKernelGGL(f16_colstats_kernel<32>, dim3(kStatBlocks), dim3(256), 0, s, X, y, n, pmax, psum);
    hipLaunchKernelGGL(f16_header_kernel<32>, dim3(1), dim3(64), 0, s, (const float*)pmax, (const double*)psum,
                       kStatBlocks, (Header*)packed);
    hipLaunchKernelGGL(f16_pack_kernel<32>, dim3(blocks), dim3(256), 0, s, X, y, n, np, (unsigned char*)packed);
  }
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_launch_linreg_f16(const void* packed, int64_t n, int d, const float* theta, int c, float* logp,
                            float* grad, float* ws, hipStream_t s) {
  if (((uintptr_t)packed & 15) != 0) return BVI_ERR_ARG;
  if (d != 32 && d != 64) return BVI_ERR_UNSUPPORTED;
  const int ncb = bvi_ceil_div(c, kChainsPerBlock);
  int per8 = 32 / ncb;
  if (per8 < 1) per8 = 1;
  const int nrb = 8 * per8;
  dim3 grid(ncb * nrb);
  const unsigned char* pk = (const unsigned char*)packed;
  const size_t lds = 2 * (size_t)(2 * kTileRows * (2 * d + 16) + kTileRows * 16);
  if (d == 64) {
    hipLaunchKernelGGL((linreg_f16_kernel<64, 4>), grid, dim3(kThreads), lds, s, pk, n, theta, c, ncb, nrb, ws);
  } else {
    hipLaunchKernelGGL((linreg_f16_kernel<32, 4>), grid, dim3(kThreads), lds, s, pk, n, theta, c, ncb, nrb, ws);
  }
  BVI_HIP_CHECK_LAUNCH();
  const int tot = c * (d + 2);
  hipLaunchKernelGGL(linreg_f16_finalize, dim3(bvi_ceil_div(tot, 256)), dim3(256), 0, s, (const float*)ws, nrb, n,
                     d, pk, theta, c, logp, grad);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}
