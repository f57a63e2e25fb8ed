// Gaussian linear-regression likelihood and gradient for every chain: exact-f32
// kernels (VALU for d <= 8, f32-input MFMA for d <= 64), the finalize step, and
// the dispatcher that also routes to the bf16-split MFMA kernel
// (linreg_bf16x3.hip).
//
// Replaces the feature-conditioned 'y' node tfd.Normal(beta_0 + X w, scale) of
// bayes_vi/model/model.py:194-195,201-203 (README.md:754-756 with the features
// as one [N,D] tensor, bayes_vi/utils/datasets.py:48-50) and its autodiff
// gradient.  Maths, per chain c with r_i = y_i - beta_0 - x_i . w:
//   S2 = sum r_i^2, S1 = sum r_i, G = sum r_i x_i            (raw sums, [d+2] per chain)
//   logp = -S2/(2 s^2) - n log s - n/2 log 2pi
//   dlogp/dbeta_0 = S1/s^2, dlogp/dw = G/s^2, dlogp/ds = -n/s + S2/s^3.
// All kernels write raw sums [row_block][chain][S2, S1, G_0..G_{d-1}] into the
// workspace; linreg_finalize adds the blocks in fixed order in double.
#include "bvi_common.h"

int bvi_launch_linreg_bf16x3(const float* X, const float* y, const void* packed, int64_t n, int d,
                             const float* theta, int c, int pieces, float* ws, int* nb_out, hipStream_t s);
int bvi_linreg_bf16x3_blocks(int64_t n, int c);

// ---------------------------------------------------------------------------
// d <= 8: VALU kernel, lane = chain + CP * row-subset (see normal_iid.hip).
// ---------------------------------------------------------------------------
template <int DP, int CP>
__global__ __launch_bounds__(256) void linreg_valu_kernel(const float* __restrict__ X,
                                                          const float* __restrict__ y, int64_t n, int d,
                                                          const float* __restrict__ theta, int C,
                                                          float* __restrict__ partials) {
  constexpr int RS = 64 / CP;
  constexpr int Q = DP + 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform by construction
  const int cl = lane & (CP - 1), rsub = lane / CP;
  const int chain = blockIdx.y * CP + cl;
  float b0 = 0.f, w[DP];
#pragma unroll
  for (int j = 0; j < DP; ++j) w[j] = 0.f;
  if (chain < C) {
    const float* th = theta + (int64_t)chain * (d + 2);
    b0 = th[0];
#pragma unroll
    for (int j = 0; j < DP; ++j)
      if (j < d) w[j] = th[1 + j];
  }
  float tot[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) tot[q] = 0.f;
  const int64_t ntiles = (n + 63) / 64;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
    if (CP == 64 && d == DP && t * 64 + 64 <= n) {
      // lane = chain: every lane needs the same 64 rows, so they come through the scalar data path
      // (wave-uniform addresses, read-only data, compile-time row stride: wide scalar loads) and reach
      // the vector ALU as scalar operands
      const float* __restrict__ yt = y + t * 64;
      const float* __restrict__ Xt = X + t * 64 * DP;
#pragma unroll 16
      for (int j = 0; j < 64; ++j) {
        float r = yt[j] - b0;
        float xj[DP];
#pragma unroll
        for (int k = 0; k < DP; ++k) {
          xj[k] = Xt[j * DP + k];
          r = fmaf(-w[k], xj[k], r);
        }
        tot[0] = fmaf(r, r, tot[0]);
        tot[1] += r;
#pragma unroll
        for (int k = 0; k < DP; ++k) tot[2 + k] = fmaf(r, xj[k], tot[2 + k]);
      }
      continue;
    }
    const int64_t i = t * 64 + lane;
    const bool ok = i < n;
    const float yv = ok ? y[i] : 0.f;
    float xv[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) xv[j] = (ok && j < d) ? X[i * d + j] : 0.f;
    const bool full = t * 64 + 64 <= n;
    float acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = 0.f;
#pragma unroll 8
    for (int j = 0; j < CP; ++j) {
      const int src = rsub + RS * j;
      float r = __shfl(yv, src, 64) - b0;
      float xj[DP];
#pragma unroll
      for (int k = 0; k < DP; ++k) {
        xj[k] = __shfl(xv[k], src, 64);
        r = fmaf(-w[k], xj[k], r);
      }
      r = (full || t * 64 + src < n) ? r : 0.f;
      acc[0] = fmaf(r, r, acc[0]);
      acc[1] += r;
#pragma unroll
      for (int k = 0; k < DP; ++k) acc[2 + k] = fmaf(r, xj[k], acc[2 + k]);
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
  const int qreal = d + 2;
  for (int idx = threadIdx.x; idx < CP * qreal; idx += 256) {
    const int c = idx / qreal, q = idx - c * qreal;
    const int gc = blockIdx.y * CP + c;
    if (gc >= C) continue;
    partials[((int64_t)blockIdx.x * C + gc) * qreal + q] = sm[0][c][q] + sm[1][c][q] + sm[2][c][q] + sm[3][c][q];
  }
}

// ---------------------------------------------------------------------------
// d in {1, 2, 4, 8}, 16-byte aligned X and y, few chains (CG = 1 or 4 per group): every lane owns groups of four
// consecutive rows -- d + 1 16-byte loads per group, two groups in flight -- and carries the sums of all CG
// chains in registers; the chains' parameters are wave-uniform (scalar registers).  No lane exchange before
// the final reduction.
// ---------------------------------------------------------------------------
template <int DP, int CG>
__global__ __launch_bounds__(256) void linreg_rows_kernel(const float* __restrict__ X, const float* __restrict__ y,
                                                          int64_t n, const float* __restrict__ theta, int C,
                                                          float* __restrict__ partials) {
  constexpr int Q = DP + 2;
  const int c0 = blockIdx.y * CG;
  float b0[CG], w[CG][DP], acc[CG][Q];
#pragma unroll
  for (int g = 0; g < CG; ++g) {
    const float* th = theta + (int64_t)(c0 + g < C ? c0 + g : 0) * (DP + 2);   // padding chains: results dropped
    b0[g] = th[0];
#pragma unroll
    for (int k = 0; k < DP; ++k) w[g][k] = th[1 + k];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[g][q] = 0.f;
  }
  auto row = [&](float yv, const float* x) {
#pragma unroll
    for (int g = 0; g < CG; ++g) {
      float r = yv - b0[g];
#pragma unroll
      for (int k = 0; k < DP; ++k) r = fmaf(-w[g][k], x[k], r);
      acc[g][0] = fmaf(r, r, acc[g][0]);
      acc[g][1] += r;
#pragma unroll
      for (int k = 0; k < DP; ++k) acc[g][2 + k] = fmaf(r, x[k], acc[g][2 + k]);
    }
  };
  auto group = [&](const f32x4& yv, const f32x4 (&xv)[DP]) {
    float flat[4 * DP];
#pragma unroll
    for (int k = 0; k < DP; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) flat[4 * k + e] = xv[k][e];
    row(yv.x, flat); row(yv.y, flat + DP); row(yv.z, flat + 2 * DP); row(yv.w, flat + 3 * DP);
  };
  const f32x4* __restrict__ X4 = reinterpret_cast<const f32x4*>(X);
  const f32x4* __restrict__ y4 = reinterpret_cast<const f32x4*>(y);
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  const int64_t n4 = n >> 2;
  int64_t i = tid;
  for (; i + nth < n4; i += 2 * nth) {
    f32x4 ya = y4[i], yb = y4[i + nth], xa[DP], xb[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) xa[k] = X4[i * DP + k];
#pragma unroll
    for (int k = 0; k < DP; ++k) xb[k] = X4[(i + nth) * DP + k];
    group(ya, xa);
    group(yb, xb);
  }
  for (; i < n4; i += nth) {
    f32x4 ya = y4[i], xa[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) xa[k] = X4[i * DP + k];
    group(ya, xa);
  }
  if (tid < n - 4 * n4) {
    const int64_t r = 4 * n4 + tid;
    float x[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) x[k] = X[r * DP + k];
    row(y[r], x);
  }
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
  for (int idx = threadIdx.x; idx < CG * Q; idx += 256) {
    const int g = idx / Q, q = idx - g * Q;
    if (c0 + g >= C) continue;
    partials[((int64_t)blockIdx.x * C + c0 + g) * Q + q] = (sm[0][g][q] + sm[1][g][q]) + (sm[2][g][q] + sm[3][g][q]);
  }
}

// ---------------------------------------------------------------------------
// 8 < d <= 64: f32-input MFMA kernel (v_mfma_f32_16x16x4_f32: bitwise an f32
// fma chain).  No LDS: each wave streams its own 16-row tiles straight into
// the two operand layouts.
//   GEMM1  R[16 rows, 16 chains] = (y - b0) - X W^T : A = X in "row" layout
//          (lane (i = l&15, kq = l>>4) holds X[i][16m+4kq+e]), B = -W with the
//          same feature permutation, 4*DT products of K = 4.
//   GEMM2  G[16 chains, 64 feats] += R^T X : the GEMM1 accumulator register j
//          of lane (chain = l&15, kq) is R[row 4kq+j][chain], exactly the A
//          operand of a K = 4 product over rows {4kq+j}; B = X in "column"
//          layout (lane (kq, nn = l&15) holds X[4kq+j][4nn+e]), 16 products.
//          Output tile e holds features 4nn+e in column nn.
// A wave owns NC chain tiles of 16; the chain group is the fastest-varying
// part of the block index so that blocks sharing rows run together.
// ---------------------------------------------------------------------------
template <bool VEC>
__device__ __forceinline__ f32x4 ld4(const float* __restrict__ X, int64_t row, int col, int d, int64_t n) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (row < n) {
    const float* p = X + row * d + col;
    if (VEC) {
      if (col + 4 <= d) v = *reinterpret_cast<const f32x4*>(p);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (col + e < d) v[e] = p[e];
    }
  }
  return v;
}

// (one chain tile per wave = the streaming regime: held to 128 registers so that four waves per SIMD keep more
// loads in flight)
template <int DT, int NC, bool VEC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NC == 1 ? 4 : 1, NC == 1 ? 4 : 8)))
void linreg_f32_kernel(const float* __restrict__ X,
                                                         const float* __restrict__ y, int64_t n, int d,
                                                         const float* __restrict__ theta, int C, int ncg,
                                                         float* __restrict__ partials) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int cg = blockIdx.x % ncg, rb = blockIdx.x / ncg, nrb = gridDim.x / ncg;
  const int cb = cg * 16 * NC;

  float wf[NC][DT][4], b0[NC];
  // branch-free: lanes of chains past the end read chain 0 and zero the values, so all loads
  // of the prologue issue back to back
#pragma unroll
  for (int t = 0; t < NC; ++t) {
    const int chain = cb + 16 * t + li;
    const float ok = chain < C ? 1.f : 0.f;
    const float* th = theta + (int64_t)(chain < C ? chain : 0) * (d + 2);
    b0[t] = ok * th[0];
#pragma unroll
    for (int m = 0; m < DT; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = 16 * m + 4 * kq + e;
        wf[t][m][e] = -ok * (f < d ? 1.f : 0.f) * th[1 + (f < d ? f : 0)];
      }
  }
  f32x4 g[NC][4];
  float s1[NC], s2[NC];
#pragma unroll
  for (int t = 0; t < NC; ++t) {
    s1[t] = 0.f;
    s2[t] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) g[t][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const int64_t ntiles = (n + 15) / 16;
  const int64_t stride = (int64_t)nrb * 4;
  int64_t t0 = (int64_t)rb * 4 + wave;
  // X is read from memory ONCE per tile, in the "column" layout (full 256-B rows, coalesced).
  // With aligned rows (VEC) the "row" layout of GEMM1 comes from a per-wave transpose through
  // LDS: 16 rows x 16 chunks of 16 B, chunk c of row r stored at c ^ r so that both the writes
  // (a row per 16 lanes) and the reads (a column of chunks per 16 lanes) are conflict-free.
  // Otherwise (odd d or unaligned X) the row layout is loaded from memory as well.
  __shared__ f32x4 tbuf[VEC ? 4 * 2 * 256 : 1];
  const bool yvec = (((uintptr_t)y) & 15) == 0;
  f32x4 xa[DT], xb[4], yv;
  // Full tiles of an aligned, 16-feature-multiple X take a branch-free path (plain 16-B loads);
  // the generic path guards every element.  The test is wave-uniform.
  const bool fast_cols = VEC && (d == 16 * DT);
  auto load_tile = [&](int64_t tile, f32x4 (&a)[DT], f32x4 (&b)[4], f32x4& yy) {
    const int64_t r0 = tile * 16;
    if (fast_cols && yvec && r0 + 16 <= n) {
      const float* base = X + (r0 + 4 * kq) * d + 4 * li;
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f32x4*>(base + j * d);
      yy = *reinterpret_cast<const f32x4*>(y + r0 + 4 * kq);
      return;
    }
    if (!VEC) {
#pragma unroll
      for (int m = 0; m < DT; ++m) a[m] = ld4<VEC>(X, r0 + li, 16 * m + 4 * kq, d, n);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = ld4<VEC>(X, r0 + 4 * kq + j, 4 * li, d, n);
#pragma unroll
    for (int j = 0; j < 4; ++j) yy[j] = (r0 + 4 * kq + j < n) ? y[r0 + 4 * kq + j] : 0.f;
  };
  if (t0 < ntiles) load_tile(t0, xa, xb, yv);
  int tb = 0;
  for (; t0 < ntiles; t0 += stride) {
    f32x4 xa_n[DT], xb_n[4], yv_n;
    const int64_t tn = t0 + stride;
    if (tn < ntiles) load_tile(tn, xa_n, xb_n, yv_n);
    if (VEC) {
      f32x4* tw = tbuf + (wave * 2 + tb) * 256;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * kq + j;
        tw[r * 16 + (li ^ r)] = xb[j];
      }
      // the wave's own LDS writes must land before its lanes read each other's chunks
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
      for (int m = 0; m < DT; ++m) xa[m] = tw[li * 16 + ((4 * m + kq) ^ li)];
      tb ^= 1;
    }
    const int64_t r0 = t0 * 16;
#pragma unroll
    for (int t = 0; t < NC; ++t) {
      f32x4 z;
#pragma unroll
      for (int j = 0; j < 4; ++j) z[j] = (r0 + 4 * kq + j < n) ? yv[j] - b0[t] : 0.f;
#pragma unroll
      for (int m = 0; m < DT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) z = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[m][e], wf[t][m][e], z, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s1[t] += z[j];
        s2[t] = fmaf(z[j], z[j], s2[t]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) g[t][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(z[j], xb[j][e], g[t][e], 0, 0, 0);
    }
    if (tn < ntiles) {
      if (!VEC) {
#pragma unroll
        for (int m = 0; m < DT; ++m) xa[m] = xa_n[m];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) xb[j] = xb_n[j];
      yv = yv_n;
    }
  }

  // cross-wave reduction through LDS, then one partial per (row block, chain)
  const int qreal = d + 2;
  __shared__ float sm[16 * NC][66];
  // waves add their tiles one after another: fixed order, so bitwise reproducible
  for (int turn = 0; turn < 4; ++turn) {
    if (wave == turn) {
#pragma unroll
      for (int t = 0; t < NC; ++t) {
        // S1/S2: lane (chain li, kq) holds the sum over its rows; add the 4 kq groups
        float a = s1[t], b = s2[t];
        a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
        if (kq == 0) {
          sm[16 * t + li][0] = (turn ? sm[16 * t + li][0] : 0.f) + b;
          sm[16 * t + li][1] = (turn ? sm[16 * t + li][1] : 0.f) + a;
        }
        // G tile e: row (chain) 4kq + j, column nn = li  <->  feature 4 li + e
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float* p = &sm[16 * t + 4 * kq + j][2 + 4 * li + e];
            *p = (turn ? *p : 0.f) + g[t][e][j];
          }
      }
    }
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < 16 * NC * qreal; idx += 256) {
    const int c = idx / qreal, q = idx - c * qreal;
    const int gc = cb + c;
    if (gc >= C) continue;
    partials[((int64_t)rb * C + gc) * qreal + q] = sm[c][q];
  }
}

// ---------------------------------------------------------------------------
// finalize: fixed-order reduction over row blocks (double) + scaling.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void linreg_finish(double s, int c, int q, int64_t n, int d, const float* __restrict__ theta,
                                              float* __restrict__ logp, float* __restrict__ grad,
                                              const double* __restrict__ ref) {
  const int q2 = d + 2;
  const double sc = (double)theta[(int64_t)c * q2 + d + 1];
  const double inv2 = 1.0 / (sc * sc), nn = (double)n;
  if (q == 0) {
    logp[c] = (float)(-0.5 * s * inv2 - nn * log(sc) - 0.5 * nn * BVI_LN2PI_D - (ref ? *ref : 0.0));
    if (grad) grad[(int64_t)c * q2 + d + 1] = (float)(-nn / sc + s * inv2 / sc);
  } else if (grad) {
    grad[(int64_t)c * q2 + (q - 1)] = (float)(s * inv2);
  }
}

// few row blocks: one thread per (chain, raw-sum index)
__global__ void linreg_finalize(const float* __restrict__ partials, int nb, int64_t n, int d,
                                const float* __restrict__ theta, int C, float* __restrict__ logp,
                                float* __restrict__ grad, const double* __restrict__ ref) {
  const int q2 = d + 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= C * q2) return;
  const int c = idx / q2, q = idx - c * q2;
  const double s = bvi_sum_partials<false>(partials + (int64_t)c * q2 + q, nb, (int64_t)C * q2);
  linreg_finish(s, c, q, n, d, theta, logp, grad, ref);
}

// many row blocks: a workgroup owns 64 consecutive outputs (consecutive lanes read consecutive sums of a row
// block); its 16 waves take every 16th row block and are combined in wave order through LDS
__global__ __launch_bounds__(1024) void linreg_finalize_tiled(const float* __restrict__ partials, int nb, int64_t n,
                                                              int d, const float* __restrict__ theta, int C,
                                                              float* __restrict__ logp, float* __restrict__ grad,
                                                              const double* __restrict__ ref) {
  const int q2 = d + 2, tot = C * q2;
  const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + o;
  __shared__ double part[16][64];
  double s = 0.0;
  if (idx < tot) {
#pragma unroll 16
    for (int b = grp; b < nb; b += 16) s += (double)partials[(int64_t)b * tot + idx];
  }
  part[grp][o] = s;
  __syncthreads();
  if (grp != 0 || idx >= tot) return;
  for (int w = 1; w < 16; ++w) s += part[w][o];
  linreg_finish(s, idx / q2, idx % q2, n, d, theta, logp, grad, ref);
}

static void launch_linreg_finalize(int tot, hipStream_t s, const float* ws, int nb, int64_t n, int d,
                                   const float* theta, int c, float* logp, float* grad, const double* ref) {
  if (nb > 32)
    hipLaunchKernelGGL(linreg_finalize_tiled, dim3(bvi_ceil_div(tot, 64)), dim3(1024), 0, s, ws, nb, n, d, theta, c,
                       logp, grad, ref);
  else
    hipLaunchKernelGGL(linreg_finalize, dim3(bvi_ceil_div(tot, 256)), dim3(256), 0, s, ws, nb, n, d, theta, c,
                       logp, grad, ref);
}

// ---------------------------------------------------------------------------
// launch policy
// ---------------------------------------------------------------------------
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static int linreg_valu_blocks(int64_t n) { return clampi(bvi_ceil_div(n, 256), 1, BVI_MAX_ROW_BLOCKS); }
static int linreg_f32_nc(int c) { return c <= 16 ? 1 : (c <= 32 ? 2 : 4); }
static int linreg_f32_blocks(int64_t n, int c) {
  const int ncg = bvi_ceil_div(c, 16 * linreg_f32_nc(c));
  // 64 rows per workgroup pass; one resident generation of workgroups (4 per CU x 256 CUs)
  return clampi(bvi_ceil_div(n, 64), 1, clampi(1024 / ncg, 1, BVI_MAX_ROW_BLOCKS));
}
// AUTO takes the bf16-split kernel only where the contraction dominates and the
// row count is large enough that its ~2^-17 per-product rounding averages out
// well below 1e-5 of the sums (the error of a sum of n terms falls as 1/sqrt(n)).
static bool use_bf16x3(int64_t n, int d, int c, int prec) {
  if (prec == BVI_PREC_F32) return false;
  const bool shape_ok = (d == 32 || d == 64);
  if (prec == BVI_PREC_BF16X3 || prec == BVI_PREC_BF16X6) return shape_ok;
  return shape_ok && c >= 128 && n >= 16384;
}

size_t bvi_ws_linreg(int64_t n, int c, int d) {
  int nb = d <= 8 ? linreg_valu_blocks(n) : linreg_f32_blocks(n, c);
  if (d == 32 || d == 64) {
    const int nb2 = bvi_linreg_bf16x3_blocks(n, c);
    nb = nb2 > nb ? nb2 : nb;
  }
  nb = nb < 769 ? 769 : nb;     // the f16-plane kernel: up to 768 row blocks + one scale per chain block
  return (size_t)nb * c * (d + 2) * sizeof(float);
}

template <int DP>
static void linreg_valu_dispatch(int cp, dim3 grid, hipStream_t s, const float* X, const float* y, int64_t n,
                                 int d, const float* theta, int c, float* ws) {
  const bool rows_ok = d == DP && (((uintptr_t)X | (uintptr_t)y) & 15) == 0;
  if (rows_ok && cp == 1) {
    hipLaunchKernelGGL((linreg_rows_kernel<DP, 1>), grid, dim3(256), 0, s, X, y, n, theta, c, ws);
    return;
  }
  if (rows_ok && cp == 4) {
    hipLaunchKernelGGL((linreg_rows_kernel<DP, 4>), grid, dim3(256), 0, s, X, y, n, theta, c, ws);
    return;
  }
  switch (cp) {
    case 1: hipLaunchKernelGGL((linreg_valu_kernel<DP, 1>), grid, dim3(256), 0, s, X, y, n, d, theta, c, ws); break;
    case 4: hipLaunchKernelGGL((linreg_valu_kernel<DP, 4>), grid, dim3(256), 0, s, X, y, n, d, theta, c, ws); break;
    case 16: hipLaunchKernelGGL((linreg_valu_kernel<DP, 16>), grid, dim3(256), 0, s, X, y, n, d, theta, c, ws); break;
    default: hipLaunchKernelGGL((linreg_valu_kernel<DP, 64>), grid, dim3(256), 0, s, X, y, n, d, theta, c, ws); break;
  }
}

template <int DT, bool VEC>
static void linreg_f32_dispatch(int nc, dim3 grid, hipStream_t s, const float* X, const float* y, int64_t n, int d,
                                const float* theta, int c, int ncg, float* ws) {
  switch (nc) {
    case 1: hipLaunchKernelGGL((linreg_f32_kernel<DT, 1, VEC>), grid, dim3(256), 0, s, X, y, n, d, theta, c, ncg, ws); break;
    case 2: hipLaunchKernelGGL((linreg_f32_kernel<DT, 2, VEC>), grid, dim3(256), 0, s, X, y, n, d, theta, c, ncg, ws); break;
    default: hipLaunchKernelGGL((linreg_f32_kernel<DT, 4, VEC>), grid, dim3(256), 0, s, X, y, n, d, theta, c, ncg, ws); break;
  }
}

int bvi_launch_linreg_packed(const void* packed, int64_t n, int d, const float* theta, int c, int prec,
                             float* logp, float* grad, float* ws, hipStream_t s, const double* logp_ref) {
  if (d != 32 && d != 64) return BVI_ERR_UNSUPPORTED;
  if (prec == BVI_PREC_F32) return BVI_ERR_UNSUPPORTED;
  int nb;
  const int st = bvi_launch_linreg_bf16x3(nullptr, nullptr, packed, n, d, theta, c, prec == BVI_PREC_BF16X3 ? 2 : 3,
                                          ws, &nb, s);
  if (st != BVI_OK) return st;
  const int tot = c * (d + 2);
  launch_linreg_finalize(tot, s, ws, nb, n, d, theta, c, logp, grad, logp_ref);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_launch_linreg(const float* X, const float* y, int64_t n, int d, const float* theta, int c, int prec,
                      float* logp, float* grad, float* ws, hipStream_t s, const double* logp_ref) {
  if (d < 1 || d > 64) return BVI_ERR_UNSUPPORTED;
  if ((prec == BVI_PREC_BF16X3 || prec == BVI_PREC_BF16X6) && !use_bf16x3(n, d, c, prec)) return BVI_ERR_UNSUPPORTED;
  int nb;
  if (use_bf16x3(n, d, c, prec)) {
    const int st = bvi_launch_linreg_bf16x3(X, y, nullptr, n, d, theta, c, prec == BVI_PREC_BF16X3 ? 2 : 3, ws, &nb, s);
    if (st != BVI_OK) return st;
  } else if (d <= 8) {
    nb = linreg_valu_blocks(n);
    const int cp = bvi_chain_pack(c);
    dim3 grid(nb, bvi_ceil_div(c, cp));
    if (d == 1) linreg_valu_dispatch<1>(cp, grid, s, X, y, n, d, theta, c, ws);
    else if (d == 2) linreg_valu_dispatch<2>(cp, grid, s, X, y, n, d, theta, c, ws);
    else if (d <= 4) linreg_valu_dispatch<4>(cp, grid, s, X, y, n, d, theta, c, ws);
    else linreg_valu_dispatch<8>(cp, grid, s, X, y, n, d, theta, c, ws);
    BVI_HIP_CHECK_LAUNCH();
  } else {
    const int nc = linreg_f32_nc(c);
    const int ncg = bvi_ceil_div(c, 16 * nc);
    nb = linreg_f32_blocks(n, c);
    dim3 grid(nb * ncg);
    const int dt = bvi_ceil_div(d, 16);
    const bool vec = (d % 4 == 0) && (((uintptr_t)X & 15) == 0);
#define BVI_F32_CASE(DT)                                                                        \
  case DT:                                                                                      \
    if (vec) linreg_f32_dispatch<DT, true>(nc, grid, s, X, y, n, d, theta, c, ncg, ws);         \
    else linreg_f32_dispatch<DT, false>(nc, grid, s, X, y, n, d, theta, c, ncg, ws);            \
    break;
    switch (dt) {
      BVI_F32_CASE(1)
      BVI_F32_CASE(2)
      BVI_F32_CASE(3)
      BVI_F32_CASE(4)
    }
#undef BVI_F32_CASE
    BVI_HIP_CHECK_LAUNCH();
  }
  const int tot = c * (d + 2);
  launch_linreg_finalize(tot, s, ws, nb, n, d, theta, c, logp, grad, logp_ref);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}
