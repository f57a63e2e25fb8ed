// Linear-regression raw sums for many chains on the f16 matrix cores.  The two contractions over all rows run
// on f16 operands of range-normalised values (11 significant bits, plus one from round-to-nearest); everything
// the f16 operands leave out is linear in the data for a given chain and is added in closed form, in double,
// from sums taken once at pack time -- so the result is that of the full-precision contraction:
//   kernel (per call, all rows):  r_k,i = y_i - b0 - xh_i . w1          residual GEMM, 1 product (xh: the f16 plane
//                                 S2_k = sum r_k^2                       of X, w1: one f16 piece of W)
//                                 G_k  = sum xh_i (rh_i + rl_i)          gradient GEMM, 2 products (R in 2 pieces:
//                                                                        its rounding residue is not linear in anything)
//   with xr = x - xh (exact in f32), dw = w - w1 (exact in f32), the true residual is r = r_k - (xr.w + xh.dw), hence
//   G  = G_k - A (w - dw) - (X^T X) dw + t1                 A = X^T X - XH^T XH,  t1 = XR^T y - b0 XR^T 1
//   S2 = S2_k - 2 (w . cr + dw . G_k)                        cr = t1 - B (w - dw),  B = XR^T XH
//        + w . A w - 2 w . B (w - dw) + dw . (X^T X - A) dw
// with X^T X, A, B (D x D) and XR^T y, XR^T 1 (D) in double behind the header of the packed dataset
// (f16_sums_kernel; A is accumulated term by term as xh_f xr_g + xr_f x_g, so nothing cancels).
// xr and dw are 2^-12-sized, so are the corrections: they are error compensation, far from the cancellation a
// Gram-matrix evaluation of the whole likelihood would suffer -- r_k differs from r by ~2^-12 of the fitted
// values, and the f32 sums of the kernel stay accurate as long as that is not far above the residuals themselves
// (signal / noise of several thousand: tests/test_gpu_parity.py).  An error in W is the same for every row, acts
// like a shifted parameter and grows with n; handled exactly this way it costs nothing per row.
// History at the headline shape (N = 10M, D = 64, 1024 chains): X, W and R in 2 pieces each with a third piece
// of W, 7 products, 6.5 ms; dw in closed form, 6 products, 5.8 ms; low plane's share of G in closed form, 5
// products, 5.1 ms; no low plane at all, 4 products, 4.25 ms; one piece of W, 3 products, 3.6 ms.  The all-bf16
// kernel (linreg_bf16x3.hip) needs 9 products on three planes.
// Same likelihood node as there: tfd.Normal(beta_0 + X w, scale)
// (bayes_vi/model/model.py:194-203, README.md:754-756).
//
// f16 has 11 significant bits but only 5 exponent bits, so every f16 operand is brought into its range by an
// exact power-of-two scale:
//   x'_id = x_id * s_d,    s_d = 2^(4 - floor(log2 max_i |x_id|))              (per column, at pack time;
//                                                                          0 for an all-zero column)
//   w''_cd = -w_cd / s_d * t_c                                              (per chain, per call)
//   t_c = 2^(14 - floor(log2 (max|y| + |b0_c| + D 2^5 max_d |w_cd / s_d|)))
// The bracket bounds |r_ic| for every row.  With the accumulator of the residual GEMM started at -t_c b0_c (a
// register constant) one fma per residual adds t_c y_i and gives the scaled residuals themselves,
//   r'_ic = t_c r_ic = t_c y_i + (-t_c b0_c + sum_d x'_id w''_cd),         |r'| < 2^15,
// with |w''| <= 2^10 / D * 64; the bound is loose by a few binades only, which leaves ~20 binades of f16 range
// below a typical residual, and the low piece of the residuals sits 2^-11 below its high piece.  S2' = sum r'^2
// and G'_cd = sum_i r'_ic x'_id are rescaled by 1 / t_c^2 and 1 / (t_c s_d) in the finalize kernel (exact: powers
// of two), which recomputes t_c with the same arithmetic.  S1 = sum_i r_i is not accumulated at all: it equals
// sum y - n b0 - w . (column sums of X), with the data sums taken once, in double, at pack time.
//
// The columns are centred at pack time: everything below that says x or X means x~ = x - shift with shift the
// column means rounded to f32 (the differences are exact in double; what their rounding to f32 loses is part of xr),
// and b0 means the intercept at the centroid, b0 + shift . w.  That intercept is orthogonal to every sum over the
// rows (sum x~ ~ 0, sum (y - ybar) ~ 0: S2 = S2_k + n db^2 up to the plane's remainders, nothing cancels), so for
// many chains it need not enter the streaming pass at all: a block of 256 chains whose natural scales lie within
// 2^8 of each other works with ONE scale T and NO intercept -- the staged targets T (y - ybar) are then the same
// numbers for every chain and are simply the accumulator the residual product starts from (no vector instruction
// per residual) -- and the finalize kernel adds the whole intercept in closed form, as it adds the remainders of W
// (db below is then b0 itself instead of a rounding remainder).  3.3 -> 3.1 - 3.2 ms at the headline shape.
// A block whose chains lie further apart (one on its way to infinity next to settled ones) keeps one scale and one
// intercept per chain and one fma per residual, as the few-chain kernel does.
//
// Packed dataset (bvi_linreg_pack, format BVI_PACK_F16), 16-B aligned throughout:
//   header [1280 B]: float colscale[64]; double colsum[64] (of x~); double ysum; float ymax, ybar; float shift[64]
//   gram   [64][64] double: X^T X (rows / columns beyond the model's features are zero)
//   low    [64][64] double A, [64][64] double Bt (Bt[g][f] = B_fg), double[64] XR^T y, double[64] XR^T 1
//   plane  [n_pad][D] f16: x'                          (D = 32 or 64 >= the model's feature count; padding
//                                                       rows and columns are zero)
//   y      [n_pad] f32                                 (0 for padding rows)
//
// Matrix instruction: v_mfma_f32_16x16x32_f16 (wave64).  A[m][k]: lane l holds m = l & 15,
// k = 8 (l >> 4) .. + 7; B[k][n]: lane l holds n = l & 15, same k; D[m][n]: lane l, register i holds
// m = 4 (l >> 4) + i, n = l & 15.  GEMM1: M = rows (A = X from LDS), N = chains (B = -W'' resident in
// registers), so a lane's accumulator registers are rows 4g + i of its chain (g = l >> 4).  Two such
// row tiles side by side are, lane for lane, the A operand of GEMM2 (M = chains, K = rows,
// N = features) with k index j <-> row 16 (j >> 2) + 4 g + (j & 3) of the 32-row pair; the B operand
// takes the same row order from two transposed LDS reads (ds_read_b64_tr_b16).
//
// The loop is bound by vector-instruction issue (an MFMA of this shape holds the issue port for 8
// of its 16 cycles), so everything per residual that is not an MFMA is kept to: one fma (start of the
// accumulator), one fma (S2'), one packed convert and one mixed-precision fma (hi / lo split).
// The chip also lowers its clock under this load (1.9 - 1.95 GHz sustained, tools/clock_probe.sh), and more so
// under a denser one: the same pass on v_mfma_f32_32x32x16_f16 (half the matrix instructions, 32 chains per wave,
// the accumulator registers 8 s .. 8 s + 7 as the A operand of the second product's k-step s) took the same number
// of cycles but ran at 1.71 - 1.75 GHz, 6.4 ms against 5.8 ms, and was dropped.
#include <stdlib.h>
#include <atomic>
#include <type_traits>
#include "bvi_common.h"

namespace {

constexpr int kTileRows = 128;             // rows staged per barrier; processed as 64-row halves
constexpr int kGramOffset = 1280;                       // X^T X, [64][64] double, behind the Header struct
constexpr int kLowOffset = kGramOffset + 64 * 64 * 8;   // A [64][64], Bt [64][64], XR^T y [64], XR^T 1 [64], double
constexpr int kLowStride = 2 * 64 * 64 + 128;
constexpr int kHeader = kLowOffset + kLowStride * 8;    // everything in front of the planes
constexpr int kMaxRowBlocks = 256;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct Header {
  float colscale[64];   // of the centred columns x - shift
  double colsum[64];    // sum_i (x_id - shift_d), exact differences summed in double
  double ysum;
  float ymax;     // max_i |y_i - ybar|
  float ybar;     // mean of y in f32: the kernels work on y - ybar and on the intercept b0 - ybar
  float shift[64];      // the column means in f32: every sum of the format is taken over x - shift
};
static_assert(sizeof(Header) <= kGramOffset, "header must fit");

// Column scale of the packed plane: |x'| = |x s_d| < 2^(kColTop + 1)
constexpr int kColTop = 4;

// w / s_d; an all-zero column (s_d = 0) takes no part in the chain's range (0, or NaN for a
// non-finite weight, as 0 * w would give)
__device__ __forceinline__ float scaled_weight(float w, float colscale) {
  // colscale is a power of two: w / colscale as one exponent shift, exact, no division and no branch
  const float q = __builtin_amdgcn_ldexpf(w, 1 - __builtin_amdgcn_frexp_expf(colscale));
  return colscale > 0.f ? q : w * 0.f;
}

__device__ __forceinline__ float col_scale(float amax) {
  // 2^(kColTop - floor(log2 amax)), kept inside the normal f32 range; 1 for zero / non-finite input
  if (!(amax > 0.f) || !isfinite(amax)) return 1.f;
  return ldexpf(1.f, min(max(kColTop - ilogbf(amax), -126), 126));
}

// The ONE power-of-two scale t of a chain: the kernel works on w'' = -(w / s) t and produces the residuals
// already scaled, r' = t r, straight out of the matrix instruction.  From the bound that holds for every row,
//   |r| <= max|y| + |b0| + D 2^(kColTop + 1) max_d |w_d / s_d|,
// t = 2^(14 - floor(log2 bound)) keeps |r'| < 2^15 and |w''| <= 2^(14 - kColTop) / D inside the f16 range.  The
// bound is loose by a few binades only, which leaves ~20 binades of f16 range below a typical residual.
__device__ __forceinline__ float chain_scale(float amax_v, float b0, float ymax, int d) {
  const float bound = ymax + fabsf(b0) + (float)d * (float)(2 << kColTop) * amax_v;
  if (!(bound > 0.f) || !isfinite(bound)) return 1.f;
  return ldexpf(1.f, min(max(14 - ilogbf(bound), -126), 126));
}

// The columns are centred at pack time (x~ = x - shift, shift = the column means in f32), so the model reads
//   y - b0 - x . w = (y - ybar) - (b0 - ybar + shift . w) - x~ . w:
// the intercept the kernels and the finalize kernel work with is the chain's fitted value at the centroid of the
// data relative to ybar.  The finalize kernel has to reproduce the kernels' float bit for bit (what the float leaves
// of the exact value is carried in closed form there), so the sum has ONE order everywhere: the 64 products
// shift_f w_f (0 beyond the model's features), added as a butterfly over the feature index (f ^ 32, 16, ..., 1),
// then (b0 - ybar) + sum, every operation rounded on its own (no contraction).
// Here: for a lane of the matrix kernels, which holds features 32 ks + 8 g + j of its chain (g = lane >> 4).
// The weights and column scales of features 32 ks + 8 g + j of one chain, all loads issued before the first is waited
// for (left to itself the compiler makes every load conditional on its feature being one of the model's and waits
// for each in turn: sixteen round trips to memory in the prologue of every workgroup).
template <int KS>
__device__ __forceinline__ void load_weight_fragment(const float* __restrict__ th, const float* __restrict__ colscale,
                                                     int dr, int g, float (&wr)[KS][8], float (&cs)[KS][8]) {
  // (colscale: any per-feature array of 64 floats in the header -- the column scales or the shifts)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = 32 * ks + 8 * g + j;
      wr[ks][j] = th[1 + (f < dr ? f : 0)];
      cs[ks][j] = colscale[f];
    }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(wr[ks][j]), "+v"(cs[ks][j]));
}

template <int KS>
__device__ __forceinline__ float centred_intercept_frag(const float* __restrict__ th,
                                                        const float* __restrict__ shift, int dr, int g, float ybar) {
  float wr[KS][8], m[KS][8];
  load_weight_fragment<KS>(th, shift, dr, g, wr, m);      // (all loads in flight together)
  float a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int f0 = 8 * g + j, f1 = 32 + 8 * g + j;
    const float p0 = f0 < dr ? __fmul_rn(m[0][j], wr[0][j]) : 0.f;
    const float p1 = (KS == 2 && f1 < dr) ? __fmul_rn(m[KS - 1][j], wr[KS - 1][j]) : 0.f;
    a[j] = __fadd_rn(p0, p1);                                   // f ^ 32
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = __fadd_rn(a[j], __shfl_xor(a[j], 32, 64));   // f ^ 16
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = __fadd_rn(a[j], __shfl_xor(a[j], 16, 64));   // f ^ 8
#pragma unroll
  for (int j = 0; j < 4; ++j) a[j] = __fadd_rn(a[j], a[j + 4]);
#pragma unroll
  for (int j = 0; j < 2; ++j) a[j] = __fadd_rn(a[j], a[j + 2]);
  return __fadd_rn(th[0] - ybar, __fadd_rn(a[0], a[1]));
}

// The same sum by a whole wave whose lane o holds feature o (the finalize kernel)
__device__ __forceinline__ float centred_intercept_wave(const float* __restrict__ th,
                                                        const float* __restrict__ shift, int d, int o, float ybar) {
  float p = o < d ? __fmul_rn(shift[o], th[1 + o]) : 0.f;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) p = __fadd_rn(p, __shfl_xor(p, off, 64));
  return __fadd_rn(th[0] - ybar, p);
}

// Scaled entry of the centred plane before its rounding to f16: the exact difference x - shift, times the column's
// power-of-two scale, rounded to float once (pack, Gram and remainder kernels all go through here)
__device__ __forceinline__ float plane_value(float x, float shift, float colscale) {
  return (float)(((double)x - (double)shift) * (double)colscale);
}

// Target value given to the padding rows of the last tile: the centred intercept of the first chain of the
// workgroup's chain block (0 if that is not finite)
__device__ __forceinline__ float pad_value(float first_chain_intercept) {
  return isfinite(first_chain_intercept) ? first_chain_intercept : 0.f;
}

// Grid of the first f16 piece of a chain's scaled weights: 2^-13 of the binade of the largest one (as a pair:
// the grid and its reciprocal, both powers of two).
__device__ __forceinline__ void piece_grid(float amax_q, float& grid, float& ginv) {
  int e = (amax_q > 0.f && isfinite(amax_q)) ? ilogbf(amax_q) - 13 : 0;
  e = min(max(e, -100), 100);
  grid = ldexpf(1.f, e);
  ginv = ldexpf(1.f, -e);
}

// First f16 piece of a scaled weight q: the nearest multiple of the chain's grid, or of the f16 spacing at q where
// that is coarser.  A weight far below its chain's largest one thus keeps only a few bits in the first piece (the rest
// goes to the compensated remainder), so that its products with x' carry no bits far below the
// ones of the large products they are summed with: the matrix instruction cuts every addend off, toward zero, a few
// bits below the largest addend's last bit, which for full-mantissa small products is a loss with a sign.
// (The matrix instruction takes f16 subnormals as they are: tools/probes/mfma_subnormal.hip.)
__device__ __forceinline__ _Float16 first_piece(float q, float grid, float ginv) {
  const float g = rintf(q * ginv) * grid;
  return fabsf(q) < 1024.f * grid ? (_Float16)g : (_Float16)q;
}

// (q0, q1) -> packed f16 (hi0, hi1) and (lo0, lo1) with lo = f16(q - hi): one packed convert and two
// mixed-precision fma that read the f16 halves directly
__device__ __forceinline__ void split_pair(float q0, float q1, unsigned& hi, unsigned& lo) {
  asm volatile(
      "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
      "v_fma_mixlo_f16 %1, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %1, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(hi), "=&v"(lo)
      : "v"(q0), "v"(q1));
}

template <int D, int NW, int CT>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(CT == 1 ? 3 : 2, CT == 1 ? 3 : 2))) void linreg_f16_kernel(
    const unsigned char* __restrict__ packed, int64_t n, int dr, const float* __restrict__ theta, int C, int ncb,
    int nrb, float* __restrict__ partials) {
  // D: feature count of the packed planes (32 or 64); dr <= D: the model's feature count (columns dr .. D-1
  // of the planes are zero and take no part)
  constexpr int KS = D / 32;   // k-steps of GEMM1
  constexpr int FT = D / 16;   // feature tiles of GEMM2
  constexpr int SA = 2 * D;    // LDS row stride: no padding, 16-B chunks XOR-swizzled by row (swz below)
  constexpr int IMG = kTileRows * SA;
  constexpr int BUF = IMG + kTileRows * 4;           // the plane, y
  constexpr int TPR = D / 8;
  constexpr int NT = 64 * NW;
  constexpr int NCH = kTileRows * TPR;               // 16-B chunks of one plane of a tile
  constexpr int PER = (NCH + NT - 1) / NT;           // chunks staged per thread and plane
  constexpr int YT = NT / kTileRows;                 // threads staging the same y
  static_assert(NT % kTileRows == 0, "y staging needs a whole number of threads per row");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 15, g = lane >> 4;
  const int x8 = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int cbi = jj % ncb, rbi = (jj / ncb) * 8 + x8;
  // CT chain tiles of 16 per wave: 2 for the many-chain shapes, 1 when at most 16 NW chains are asked for
  const int cbase = cbi * (16 * CT * NW) + 16 * CT * wave;
  const bool wave_has_chains = __builtin_amdgcn_readfirstlane(cbase) < C;      // wave-uniform, in a scalar register
  const Header* hdr = reinterpret_cast<const Header*>(packed);
  const int64_t ntiles = (n + kTileRows - 1) / kTileRows;
  const int64_t n_pad = ntiles * kTileRows;
  const unsigned char* planes = packed + kHeader;
  const unsigned char* yplane = planes + (int64_t)n_pad * D * 2;

  // scaled -W fragments (one f16 piece); per chain: its scale t and nb0t = -t b0
  f16x8 w1[CT][KS];
  float tc[CT];
  f32x4v nb0v[CT];
  {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int chain = cbase + 16 * ct + ln;
      const bool cvalid = chain < C;
      const float* th = theta + (int64_t)(cvalid ? chain : 0) * (dr + 2);
      float v[KS][8], amax = 0.f;
      float wr[KS][8], cs[KS][8];
      load_weight_fragment<KS>(th, hdr->colscale, dr, g, wr, cs);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int f = 32 * ks + 8 * g + j;
          v[ks][j] = (cvalid && f < dr) ? -scaled_weight(wr[ks][j], cs[ks][j]) : 0.f;
          amax = fmaxf(amax, fabsf(v[ks][j]));
        }
      amax = fmaxf(amax, __shfl_xor(amax, 16, 64));   // the four lane groups hold the chain's D features
      amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
      // the intercept against the centred targets and columns
      const float b0i = centred_intercept_frag<KS>(th, hdr->shift, dr, g, hdr->ybar);
      const float b0 = cvalid ? b0i : 0.f;
      const float t = chain_scale(amax, b0, hdr->ymax, D);
      float grid, ginv;
      piece_grid(amax * t, grid, ginv);
      tc[ct] = t;
#pragma unroll
      for (int i = 0; i < 4; ++i) nb0v[ct][i] = -b0 * t;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          w1[ct][ks][j] = first_piece(v[ks][j] * t, grid, ginv);   // what is left, q - w1, is accounted for in the finalize kernel
    }
  }

  f32x4v G[CT][FT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int i = 0; i < 4; ++i) G[ct][ft][i] = 0.f;
  // per-tile sums of squares are floats, their running total a double: at 10M rows a chain's S2' is ~1e7 times a
  // typical square, and the log-density differences a sampler decides on must come out to ~1e-3 of one square
  double s2[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) s2[ct] = 0.0;

  // chunk q of row r lives at chunk q ^ swz(r): with it the A-operand ds_read_b128 (lane groups
  // {0-3,12-15,20-27}, ...: 16 rows, two neighbouring chunks), the transposed ds_read_b64_tr_b16 (32 lanes:
  // 8 rows x 32 B) and the 128-bit row stores all spread over the 64 banks without conflicts
  auto swz = [](int row) { return D == 64 ? (row & 7) : ((0x78 >> (2 * ((row >> 2) & 3))) & 3); };
  // staged through registers two tiles ahead (two register sets, taken in turn): at few chains a tile's arithmetic is
  // a fraction of the HBM latency, and with one tile in flight per workgroup the loads' latency, not the memory
  // system's pace, set the time
  s16x8 xp[2][PER];
  float yp[2];
  bool yvalid[2];
  const float ybar = hdr->ybar;
  const float ypad = pad_value(centred_intercept_frag<KS>(theta + (int64_t)cbi * (16 * CT * NW) * (dr + 2), hdr->shift, dr, g, ybar));
  auto gload = [&](int64_t tile, auto set_c) {
    constexpr int S = decltype(set_c)::value;
    const int64_t row0 = (tile < ntiles ? tile : 0) * kTileRows;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int idx = tid + e * NT;
      const int srow = (NCH % NT == 0 || idx < NCH) ? idx / TPR : 0, ch = idx % TPR;
      xp[S][e] = *reinterpret_cast<const s16x8*>(planes + ((row0 + srow) * D + ch * 8) * 2);
    }
    // every thread stages a y (NW threads per row, same value): an unconditional use keeps the
    // compiler from sinking this load down to the store, where its latency would be exposed
    // padding rows (x = 0): G' gains nothing from them; with the target set to the centred intercept of the
    // workgroup's first chain their residuals, t (ypad - (b0 - ybar)), are zero for that chain and of the size of
    // the chains' spread for the others -- small next to the real rows' sum -- and the finalize kernel takes
    // their squares out of S2' again
    yp[S] = *reinterpret_cast<const float*>(yplane + (row0 + tid / YT) * 4);
    yvalid[S] = row0 + tid / YT < n;      // (the arithmetic on yp waits until the store: no wait for the load here)
  };
  auto lstore = [&](int buf, auto set_c) {
    constexpr int S = decltype(set_c)::value;
    unsigned char* base = lds + buf * BUF;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int idx = tid + e * NT;
      const int srow = idx / TPR, ch = idx % TPR;
      if (NCH % NT == 0 || idx < NCH)
        *reinterpret_cast<s16x8*>(base + srow * SA + ((ch ^ swz(srow)) << 4)) = xp[S][e];
    }
    *reinterpret_cast<float*>(base + IMG + (tid / YT) * 4) = yvalid[S] ? yp[S] - ybar : ypad;
  };

  // slot behind the partials that tells the finalize kernel how this chain block was scaled: 0 = one scale per chain
  if (rbi == 0 && tid == 0) partials[(int64_t)nrb * C * (dr + 2) + cbi] = 0.f;
  int64_t tile = rbi;
  constexpr std::integral_constant<int, 0> set0{};
  constexpr std::integral_constant<int, 1> set1{};
  if (tile < ntiles) {
    gload(tile, set0);
    lstore(0, set0);
    gload(tile + nrb, set1);      // past the end: re-reads tile 0, never used
  }
  __syncthreads();
  int buf = 0;
  // iteration k (parity P = k % 2): loads tile k + 2 into register set P, multiplies tile k from LDS, stores tile k + 1
  // (register set 1 - P, loaded during iteration k - 1) into the other LDS buffer
  auto iteration = [&](auto par_c) {
    constexpr int P = decltype(par_c)::value;
    gload(tile + 2 * (int64_t)nrb, std::integral_constant<int, P>{});
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* base = lds + buf * BUF;
    float a2[CT];               // this tile's sum of squares: two-level summation keeps S2 to ~1e-7
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) a2[ct] = 0.f;
    // a wave whose chain tiles all lie beyond the last chain only helps with the staging (with 1 - 16 chains three of
    // the four waves: the tile then costs a quarter of the matrix and vector work and the pass runs at the memory
    // system's pace)
    if (wave_has_chains)
#pragma unroll
    for (int sub = 0; sub < kTileRows / 64; ++sub) {
    const int ro = 64 * sub;
    f32x4v Z[4][CT];
    // LDS fragments are fetched one step ahead of the MFMAs that consume them (the scheduler
    // barriers keep the fetches where they are written), so that their latency is not exposed
    f16x8 ah[2][KS];
    auto load_a = [&](int rt, int slot) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int off = (ro + 16 * rt + ln) * SA + (((4 * ks + g) ^ swz(16 * rt + ln)) << 4);
        ah[slot][ks] = *reinterpret_cast<const f16x8*>(base + off);
      }
    };
    f16x8 xh[3];
    auto load_x = [&](int m, int ft, int slot) {
      const int trow = ro + 32 * m + 4 * g + (ln >> 2);   // swz(trow + 16) == swz(trow)
      const int off0 = trow * SA + (((2 * ft + ((ln & 3) >> 1)) ^ swz(trow)) << 4) + 8 * (ln & 1);
      const int off1 = off0 + 16 * SA;
      const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off0));
      const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off1));
      xh[slot] = __builtin_bit_cast(f16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // GEMM1 on top of -t b0, plus t y: the scaled residuals r' = t (y - b0 - xh . w1) of this wave's 64 rows x 32 chains
    load_a(0, 0);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const f32x4v yv = *reinterpret_cast<const f32x4v*>(base + IMG + (ro + 16 * rt + 4 * g) * 4);
      if (rt < 3) load_a(rt + 1, (rt + 1) & 1);
      else { load_x(0, 0, 0); load_x(0, 1, 1); }
      __builtin_amdgcn_sched_barrier(0);
      // the accumulator starts at -t b0 (a register constant: no instruction), t y is added afterwards in f32:
      // inside the matrix instruction the large t y would set the exponent every product is cut off at
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          Z[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt & 1][ks], w1[ct][ks], ks == 0 ? nb0v[ct] : Z[rt][ct],
                                                             0, 0, 0);
        }
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) Z[rt][ct][i] = fmaf(yv[i], tc[ct], Z[rt][ct][i]);
    }
    // GEMM2: G' += R'^T Xh' over the two 32-row pairs (the low plane's share is added in closed form by the
    // finalize kernel)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      f16x8 rh[CT], rl[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        unsigned hw[4], lw[4];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const float q0 = Z[2 * m + (j >> 2)][ct][j & 3], q1 = Z[2 * m + (j >> 2)][ct][(j & 3) + 1];
          a2[ct] = fmaf(q0, q0, a2[ct]);
          a2[ct] = fmaf(q1, q1, a2[ct]);
          split_pair(q0, q1, hw[j >> 1], lw[j >> 1]);
        }
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        rh[ct] = __builtin_bit_cast(f16x8, (u32x4){hw[0], hw[1], hw[2], hw[3]});
        rl[ct] = __builtin_bit_cast(f16x8, (u32x4){lw[0], lw[1], lw[2], lw[3]});
      }
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) {
        const int step = m * FT + ft, cur = step % 3;     // fragments are fetched two steps ahead
        if (step + 2 < 2 * FT) load_x((step + 2) / FT, (step + 2) % FT, (step + 2) % 3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          G[ct][ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh[ct], xh[cur], G[ct][ft], 0, 0, 0);
          G[ct][ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rl[ct], xh[cur], G[ct][ft], 0, 0, 0);
        }
      }
    }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) s2[ct] += (double)a2[ct];
    lstore(buf ^ 1, std::integral_constant<int, 1 - P>{});
    __syncthreads();
    buf ^= 1;
  };
  while (tile < ntiles) {
    iteration(set0);
    tile += nrb;
    if (tile >= ntiles) break;
    iteration(set1);
    tile += nrb;
  }

#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    double v = s2[ct];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    const int chain = cbase + 16 * ct + ln;
    if (chain < C && g == 0) {
      // S2' as a float pair (high part, remainder); S1 comes from the packed data sums (finalize)
      float* p = partials + ((int64_t)rbi * C + chain) * (dr + 2);
      p[0] = (float)v;
      p[1] = (float)(v - (double)(float)v);
    }
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = cbase + 16 * ct + 4 * g + i;
        if (c < C && 16 * ft + ln < dr) partials[((int64_t)rbi * C + c) * (dr + 2) + 2 + 16 * ft + ln] = G[ct][ft][i];
      }
  }
}

// ---- the same pass, software-pipelined (many chains: two 16-chain tiles per wave) ----
// The loop above runs its three stages one after the other -- residual GEMM (matrix + vector work), hi / lo split
// (vector work only), gradient GEMM (matrix work only) -- so a wave leaves the matrix pipe idle while it splits and
// the vector ALU idle while it multiplies, and the two waves of a SIMD, released by the same barrier, do so at the
// same time; LDS fragments are fetched one short step ahead and their latency shows.  Here the tile is cut into 32-row
// UNITS and three consecutive units are in flight per wave.  In block j the wave issues, instruction by instruction
// interleaved,
//     R(j-1): the gradient GEMM of the unit whose residuals were split in the previous block      (2 FT CT MFMAs)
//     Q(j)  : + t y, sum of squares and hi / lo split of the unit the previous block multiplied   (56 vector ops)
//     P(j+1): the residual GEMM of the next unit                                                  (2 KS CT MFMAs)
// and the LDS reads of everything block j+1 will consume -- a whole block (~400 cycles) ahead.  The three streams are
// independent inside a block, so the order below is fixed by hand: per group of three slots two R MFMAs and one P MFMA,
// each followed by the 2 - 3 vector instructions that fit into the 8 issue cycles an MFMA of this shape leaves free.
// Units cross tile boundaries (the last units of a tile are multiplied while the next tile's first unit is read), so
// the staged tiles live in a ring of three LDS buffers, filled two tiles ahead; still one barrier per tile.
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// GRAD = false (value only: what random-walk Metropolis asks for).  S2's compensation still needs G_k = sum xh r_k,
// but only through the product dw . G_k with dw ~ 2^-13 of the weights, for which the high f16 piece of the
// residuals is plenty: the low piece, its share of the split and its gradient products are left out (a third of
// the matrix instructions and two of seven vector instructions per residual pair).
template <int D, int NW, bool GRAD>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void linreg_f16_pipe_kernel(
    const unsigned char* __restrict__ packed, int64_t n, int dr, const float* __restrict__ theta, int C, int ncb,
    int nrb, float* __restrict__ partials) {
  constexpr int CT = 2;
  constexpr int KS = D / 32;   // k-steps of the residual GEMM
  constexpr int FT = D / 16;   // feature tiles of the gradient GEMM
  constexpr int SA = 2 * D;    // LDS row stride (16-B chunks XOR-swizzled by row, as above)
  constexpr int IMG = kTileRows * SA;
  constexpr int BUF = IMG + kTileRows * 4;
  constexpr int RING = 3;
  constexpr int TPR = D / 8;
  constexpr int NT = 64 * NW;
  constexpr int NCH = kTileRows * TPR;
  constexpr int PER = (NCH + NT - 1) / NT;
  constexpr int YT = NT / kTileRows;
  constexpr int UPT = kTileRows / 32;          // units per tile
  constexpr int NP = 2 * CT * KS;              // P MFMAs per block
  constexpr int NS = 3 * NP;                   // slots (= MFMAs) per block: two R per P
  constexpr int NG = NS / 3;                   // groups of three slots
  constexpr int JPG = 4 * CT / NG;             // hi / lo jobs (pairs of residuals of one chain tile) per group
  static_assert(2 * FT * CT == 2 * NP && NT % kTileRows == 0 && UPT == 4 && (4 * CT) % NG == 0, "schedule shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool lead_wave = __builtin_amdgcn_readfirstlane(wave) < NW / 2;      // wave-uniform, in a scalar register
  const int ln = lane & 15, g = lane >> 4;
  const int x8 = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int cbi = jj % ncb, rbi = (jj / ncb) * 8 + x8;
  const int cbase = cbi * (16 * CT * NW) + 16 * CT * wave;
  // a wave whose chain tiles all lie beyond the last chain (a partly filled chain block) only helps with the staging
  const bool wave_has_chains = __builtin_amdgcn_readfirstlane(cbase) < C;
  const Header* hdr = reinterpret_cast<const Header*>(packed);
  const int64_t ntiles = (n + kTileRows - 1) / kTileRows;
  const int64_t n_pad = ntiles * kTileRows;
  const unsigned char* planes = packed + kHeader;
  const unsigned char* yplane = planes + (int64_t)n_pad * D * 2;

  // Scales.  Each chain has its natural power-of-two scale t_c (chain_scale: the largest that keeps its scaled
  // residuals inside the f16 range).  When the natural scales of the workgroup's chains lie within 2^8 of each other
  // -- chains of one run do -- the whole block works with ONE scale T, the smallest of them, and with no intercept:
  // the staged targets are T (y - ybar), the same numbers for every chain, and enter the residual product as its
  // accumulator (no vector instruction per residual); with the columns centred at pack time the intercept
  // b0 - ybar + shift . w is orthogonal to everything the pass sums (S2 = S2_k + n db^2 up to the plane's remainders:
  // nothing cancels) and the finalize kernel adds it in closed form, as it adds the remainders of W.  A block whose
  // chains are further apart (a chain on its way to infinity next to settled ones) keeps one scale and one
  // intercept per chain and pays one fma per residual for the targets.
  __shared__ float sred[2 * NW];
  f16x8 w1[CT][KS];
  float tc[CT];
  f32x4v nb0v[CT];
  float T;
  bool uni;
  {
    float v[CT][KS][8], amaxc[CT], b0c[CT];
    float tlo = INFINITY, thi = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int chain = cbase + 16 * ct + ln;
      const bool cvalid = chain < C;
      const float* th = theta + (int64_t)(cvalid ? chain : 0) * (dr + 2);
      float amax = 0.f;
      float wr[KS][8], cs[KS][8];
      load_weight_fragment<KS>(th, hdr->colscale, dr, g, wr, cs);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int f = 32 * ks + 8 * g + j;
          v[ct][ks][j] = (cvalid && f < dr) ? -scaled_weight(wr[ks][j], cs[ks][j]) : 0.f;
          amax = fmaxf(amax, fabsf(v[ct][ks][j]));
        }
      amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
      amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
      amaxc[ct] = amax;
      const float b0i = centred_intercept_frag<KS>(th, hdr->shift, dr, g, hdr->ybar);
      b0c[ct] = cvalid ? b0i : 0.f;
      // the chain's scale without an intercept; chains whose bound is not finite (they return NaN whatever the
      // scale) and lanes beyond the last chain take no part in the block's choice
      const float bound = hdr->ymax + (float)D * (float)(2 << kColTop) * amax;
      if (cvalid && bound > 0.f && isfinite(bound)) {
        const float tn = chain_scale(amax, 0.f, hdr->ymax, D);
        tlo = fminf(tlo, tn);
        thi = fmaxf(thi, tn);
      }
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      tlo = fminf(tlo, __shfl_xor(tlo, off, 64));
      thi = fmaxf(thi, __shfl_xor(thi, off, 64));
    }
    if (lane == 0) { sred[wave] = tlo; sred[NW + wave] = thi; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) { tlo = fminf(tlo, sred[w]); thi = fmaxf(thi, sred[NW + w]); }
    T = tlo < INFINITY ? tlo : 1.f;
    uni = !(thi > T * 256.f);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const float t = uni ? T : chain_scale(amaxc[ct], b0c[ct], hdr->ymax, D);
      float grid, ginv;
      piece_grid(amaxc[ct] * t, grid, ginv);
      tc[ct] = t;
#pragma unroll
      for (int i = 0; i < 4; ++i) nb0v[ct][i] = -b0c[ct] * t;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) w1[ct][ks][j] = first_piece(v[ct][ks][j] * t, grid, ginv);
    }
  }
  // tells the finalize kernel which of the two this block did (slot behind the partials)
  if (rbi == 0 && tid == 0) partials[(int64_t)nrb * C * (dr + 2) + cbi] = uni ? T : 0.f;

  f32x4v G[CT][FT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int i = 0; i < 4; ++i) G[ct][ft][i] = 0.f;
  double s2[CT] = {0.0, 0.0};   // running total of the per-block float sums a2 (see linreg_f16_kernel)
  float a2[CT] = {0.f, 0.f};

  // tiles of this row block: rbi, rbi + nrb, ...
  const int64_t nt = rbi < ntiles ? (ntiles - rbi + nrb - 1) / nrb : 0;

  // ---- staging: global -> registers -> LDS ring ----
  auto swz = [](int row) { return D == 64 ? (row & 7) : ((0x78 >> (2 * ((row >> 2) & 3))) & 3); };
  s16x8 xp[PER];
  float yp;
  bool yvalid;
  const float ybar = hdr->ybar;
  // one scale per block: the staged targets are T (y - ybar), padding rows exact zeros
  const float ypad = uni ? 0.f : pad_value(centred_intercept_frag<KS>(theta + (int64_t)cbi * (16 * CT * NW) * (dr + 2), hdr->shift, dr, g, ybar));
  const float ymul = uni ? T : 1.f;
  // per-thread byte offsets inside a tile (32 bits, fixed for the whole launch); the tile's own base is wave-uniform
  // and stays in scalar registers, so a load is one instruction with no 64-bit vector arithmetic in front of it
  unsigned xoff[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int idx = tid + e * NT;
    const int srow = (NCH % NT == 0 || idx < NCH) ? idx / TPR : 0, ch = idx % TPR;
    xoff[e] = (unsigned)((srow * D + ch * 8) * 2);
  }
  const unsigned yoff = (unsigned)((tid / YT) * 4);
  auto gload = [&](int64_t k) {      // k-th tile of this row block; past the end: tile 0 of the dataset, never used
    const int64_t tile = rbi + k * nrb;
    const int64_t row0 = (tile < ntiles ? tile : 0) * kTileRows;
    const unsigned char* xbase = planes + row0 * (D * 2);
    const unsigned char* ybase = yplane + row0 * 4;
#pragma unroll
    for (int e = 0; e < PER; ++e) xp[e] = *reinterpret_cast<const s16x8*>(xbase + xoff[e]);
    yp = *reinterpret_cast<const float*>(ybase + yoff);
    yvalid = row0 + tid / YT < n;      // (the arithmetic on yp waits until the store: no wait for the load here)
  };
  auto lstore = [&](int ring) {
    unsigned char* base = lds + ring * BUF;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int idx = tid + e * NT;
      const int srow = idx / TPR, ch = idx % TPR;
      if (NCH % NT == 0 || idx < NCH)
        *reinterpret_cast<s16x8*>(base + srow * SA + ((ch ^ swz(srow)) << 4)) = xp[e];
    }
    *reinterpret_cast<float*>(base + IMG + (tid / YT) * 4) = yvalid ? (yp - ybar) * ymul : ypad;
  };

  // ---- per-lane LDS offsets of the three kinds of fragment inside a tile image (unit 0) ----
  int offA[KS], offX[FT];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) offA[ks] = ln * SA + (((4 * ks + g) ^ swz(ln)) << 4);
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) {
    const int trow = 4 * g + (ln >> 2);       // swz(trow + 16) == swz(trow)
    offX[ft] = trow * SA + (((2 * ft + ((ln & 3) >> 1)) ^ swz(trow)) << 4) + 8 * (ln & 1);
  }
  const int offY = IMG + 16 * g;

  // fragment sets, indexed by block parity (all indices are compile-time constants: registers)
  f32x4v Z[2][2][CT];          // scaled residuals of a unit: [set][row tile][chain tile]
  unsigned RH[2][CT][4], RL[2][CT][4];
  s16x4 XH[2][FT][2];          // the two 16-row halves of a transposed fragment
  f16x8 A[2][2][KS];
  f32x4v YV[2][2];
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  auto read_a = [&](const unsigned char* img, int unit, int rt, int ks) {
    return *reinterpret_cast<const f16x8*>(img + (32 * unit + 16 * rt) * SA + offA[ks]);
  };
  auto read_y = [&](const unsigned char* img, int unit, int rt) {
    return *reinterpret_cast<const f32x4v*>(img + (32 * unit + 16 * rt) * 4 + offY);
  };
  auto read_x_half = [&](const unsigned char* img, int unit, int ft, int half) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + (32 * unit + 16 * half) * SA + offX[ft]));
  };

  // UNI: the staged targets are the accumulator the residual product starts from (read a block earlier, no fma);
  // otherwise it starts from -t b0 and the targets are added by one fma per residual
  auto pass = [&](auto uni_c) {
  constexpr bool UNI = decltype(uni_c)::value;
  if (nt > 0) {
    // prologue: tiles 0 and 1 into the ring; P of unit 0; fragments of what block 0 consumes
    gload(0); lstore(0);
    gload(1); lstore(1);
    __syncthreads();
    if (wave_has_chains) {
      const unsigned char* img = lds;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        f16x8 a0[KS];
        YV[0][rt] = read_y(img, 0, rt);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a0[ks] = read_a(img, 0, rt, ks);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            Z[0][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                a0[ks], w1[ct][ks], ks == 0 ? (UNI ? YV[0][rt] : nb0v[ct]) : Z[0][rt][ct], 0, 0, 0);
        if constexpr (UNI) YV[1][rt] = read_y(img, 1, rt);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) A[1][rt][ks] = read_a(img, 1, rt, ks);
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) { RH[1][ct][i] = 0u; RL[1][ct][i] = 0u; }
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int h = 0; h < 2; ++h) XH[1][ft][h] = (s16x4){0, 0, 0, 0};
    }

    int rcur = 0;                                  // ring slot of the tile being processed
    for (int64_t k = 0; k < nt; ++k) {
      gload(k + 2);
      const int rnxt = rcur == RING - 1 ? 0 : rcur + 1;
      const unsigned char* icur = lds + rcur * BUF;
      const unsigned char* inxt = lds + rnxt * BUF;
      __builtin_amdgcn_sched_barrier(0);
      if (wave_has_chains)
      static_for<UPT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int sq = j & 1, sr = (j + 1) & 1;      // sets: Q / loads of XH, A  |  R, P, loads of YV
        // unit whose A fragments are read (for the P of the next block) and whose y is read (for its Q)
        // (UNI: y a block earlier, it is an operand of the next block's P)
        constexpr int ua = (j + 2) % UPT, uy = (j + (UNI ? 2 : 1)) % UPT;
        // The two waves of a SIMD (w and w + NW / 2) run the same stream; left alone the older one wins every issue
        // slot, reaches the tile's barrier early and waits there while the other finishes on its own.  Handing the
        // priority over in the middle of the tile lets the second one catch up before the barrier (1.8 % against no
        // priorities at all; every other pattern of the four units measured gained less or lost, three units to one
        // 15 %).  8-wave workgroups only: with 4 waves the SIMD's other wave belongs to another workgroup.  The
        // condition has to sit in a scalar register: on a per-lane condition both s_setprio are executed by every wave.
        if constexpr (NW == 8 && (j == 0 || j == UPT / 2)) bvi_set_priority(lead_wave == (j == 0));
        const unsigned char* ia = j + 2 < UPT ? icur : inxt;
        const unsigned char* iy = j + (UNI ? 2 : 1) < UPT ? icur : inxt;
        static_for<NS>([&](auto sc_) {
          constexpr int s = decltype(sc_)::value;
          constexpr int grp = s / 3, r = s % 3;
          // -- LDS reads for the next block
          constexpr int xslots = NS / FT, aslots = NS / (2 * KS);
          if constexpr (s % xslots < 2) {
            constexpr int ft = s / xslots, half = s % xslots;
            XH[sq][ft][half] = read_x_half(icur, j, ft, half);
          }
          if constexpr (s % aslots == 2) {
            constexpr int i = s / aslots, rt = i / KS, ks = i % KS;
            A[sq][rt][ks] = read_a(ia, ua, rt, ks);
          }
          if constexpr (s == 3 || s == 9) {
            constexpr int rt = s == 3 ? 0 : 1;
            YV[UNI ? sq : sr][rt] = read_y(iy, uy, rt);
          }
          // -- the slot's MFMA
          constexpr int kPSlot = 0;                        // the group's P first, then its two R (0.4 % over R, R, P)
          if constexpr (r == kPSlot) {                     // P(j+1): row tile rt, k-step ks, chain tile ct
            constexpr int rt = grp / (CT * KS), ks = (grp / CT) % KS, ct = grp % CT;
            Z[sr][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                A[sr][rt][ks], w1[ct][ks], ks == 0 ? (UNI ? YV[sr][rt] : nb0v[ct]) : Z[sr][rt][ct], 0, 0, 0);
          } else {                                         // R(j-1): feature tile ft, piece, chain tile ct
            constexpr int rr = r - (r > kPSlot ? 1 : 0);
            constexpr int q = 2 * grp + rr, ft = q / (2 * CT), piece = (q % (2 * CT)) / CT, ct = q % CT;
            const f16x8 rv = piece == 0
                ? __builtin_bit_cast(f16x8, (u32x4){RH[sr][ct][0], RH[sr][ct][1], RH[sr][ct][2], RH[sr][ct][3]})
                : __builtin_bit_cast(f16x8, (u32x4){RL[sr][ct][0], RL[sr][ct][1], RL[sr][ct][2], RL[sr][ct][3]});
            const f16x8 xv = __builtin_bit_cast(
                f16x8, __builtin_shufflevector(XH[sr][ft][0], XH[sr][ft][1], 0, 1, 2, 3, 4, 5, 6, 7));
            if constexpr (GRAD || piece == 0) G[ct][ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rv, xv, G[ct][ft], 0, 0, 0);
          }
          // -- the slot's share of Q(j): job = (chain tile ct, residual pair jp) -> element pair of Z[sq]
          static_for<JPG>([&](auto jb_) {
            constexpr int job = grp * JPG + decltype(jb_)::value;      // 0 .. 4 CT - 1, row tile 0 first
            constexpr int jp = job / CT, ct = job % CT;                // pair jp: k = 2 jp, 2 jp + 1 of the A fragment
            constexpr int rt = jp >> 1, i0 = 2 * (jp & 1);
            // balanced over the group's three MFMA gaps by issue cost (the mixed-precision fma costs ~2.6 plain ones):
            // [+ t y, + t y, convert]  [square, low piece of the first]  [square, low piece of the second]
            if constexpr (r == 0) {
              if constexpr (!UNI) {
                Z[sq][rt][ct][i0] = fmaf(YV[sq][rt][i0], tc[ct], Z[sq][rt][ct][i0]);
                Z[sq][rt][ct][i0 + 1] = fmaf(YV[sq][rt][i0 + 1], tc[ct], Z[sq][rt][ct][i0 + 1]);
              }
              const float q0 = Z[sq][rt][ct][i0], q1 = Z[sq][rt][ct][i0 + 1];
              unsigned hi;
              asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(q0), "v"(q1));
              RH[sq][ct][jp] = hi;
              if constexpr (UNI) a2[ct] = fmaf(q0, q0, a2[ct]);     // (no target fma here: the squares move up a gap)
            } else if constexpr (r == 1) {
              const float q0 = Z[sq][rt][ct][i0];
              const unsigned hi = RH[sq][ct][jp];
              unsigned lo;
              if constexpr (UNI) {
                const float q1 = Z[sq][rt][ct][i0 + 1];
                a2[ct] = fmaf(q1, q1, a2[ct]);
              } else {
                a2[ct] = fmaf(q0, q0, a2[ct]);
              }
              if constexpr (GRAD) {
                asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(q0), "v"(hi));
                RL[sq][ct][jp] = lo;
              }
            } else {
              const float q1 = Z[sq][rt][ct][i0 + 1];
              const unsigned hi = RH[sq][ct][jp];
              unsigned lo = RL[sq][ct][jp];
              if constexpr (!UNI) a2[ct] = fmaf(q1, q1, a2[ct]);
              if constexpr (GRAD) {
                asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(q1), "v"(hi));
                RL[sq][ct][jp] = lo;
              }
            }
          });
          __builtin_amdgcn_sched_barrier(0);
        });
      });
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) { s2[ct] += (double)a2[ct]; a2[ct] = 0.f; }
      lstore(rnxt == RING - 1 ? 0 : rnxt + 1);
      __syncthreads();
      rcur = rnxt;
    }
    // epilogue: the gradient GEMM of the last unit
    if (wave_has_chains)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const f16x8 rh = __builtin_bit_cast(f16x8, (u32x4){RH[1][ct][0], RH[1][ct][1], RH[1][ct][2], RH[1][ct][3]});
        const f16x8 rl = __builtin_bit_cast(f16x8, (u32x4){RL[1][ct][0], RL[1][ct][1], RL[1][ct][2], RL[1][ct][3]});
        const f16x8 xv = __builtin_bit_cast(f16x8, __builtin_shufflevector(XH[1][ft][0], XH[1][ft][1], 0, 1, 2, 3, 4, 5, 6, 7));
        G[ct][ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh, xv, G[ct][ft], 0, 0, 0);
        if constexpr (GRAD) G[ct][ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rl, xv, G[ct][ft], 0, 0, 0);
      }
  }
  };
  if (uni) pass(std::true_type{});
  else pass(std::false_type{});

#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    double v = s2[ct];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    const int chain = cbase + 16 * ct + ln;
    if (chain < C && g == 0) {
      float* p = partials + ((int64_t)rbi * C + chain) * (dr + 2);
      p[0] = (float)v;
      p[1] = (float)(v - (double)(float)v);
    }
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = cbase + 16 * ct + 4 * g + i;
        if (c < C && 16 * ft + ln < dr) partials[((int64_t)rbi * C + c) * (dr + 2) + 2 + 16 * ft + ln] = G[ct][ft][i];
      }
  }
}

// ---- pack: column statistics (two-stage, fixed order) then the split ----
// stage 1: block b reduces rows b, b + nblk, ... : per column sum (double), minimum and maximum; sum, minimum and
// maximum of y.  One pass gives everything the header needs: the column means are the shifts, and for the centred
// columns x - shift the largest |x - shift| is taken at the column's minimum or maximum (rounding to float is
// monotone) and their sum is sum(x) - n shift in double.  (The first version read the matrix a second time for
// those two.)
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void f16_colstats_kernel(
    const float* __restrict__ X, const float* __restrict__ y,
                                                           int64_t n, int dr, float* __restrict__ pmax,
                                                           double* __restrict__ psum) {
  // 256 threads = RL row lanes x D / 4 column groups: a thread owns four neighbouring columns (one 16-byte load per
  // row when the rows are D floats long and 16-byte aligned; element loads otherwise), four rows in flight
  constexpr int CGR = D / 4, RL = 256 / CGR;
  const int cg = threadIdx.x % CGR, rl = threadIdx.x / CGR, c0 = 4 * cg;
  float mn[4], mx[4];                         // (NaN: kept out by fmaxf / fminf, found by the sums)
#pragma unroll
  for (int j = 0; j < 4; ++j) { mn[j] = INFINITY; mx[j] = -INFINITY; }
  float yhi = -INFINITY, ylo = INFINITY;
  double sum[4] = {0.0, 0.0, 0.0, 0.0}, ysum = 0.0;
  const bool vec = dr == D && (((uintptr_t)X & 15) == 0);
  const int64_t stride = (int64_t)gridDim.x * RL;
  auto take = [&](const float (&v)[4], int64_t row) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x = c0 + j < dr ? v[j] : 0.f;
      mn[j] = fminf(mn[j], x);
      mx[j] = fmaxf(mx[j], x);
      sum[j] += (double)x;
    }
    if (cg == 0) {
      const float yv = y[row];
      ysum += (double)yv;
      yhi = fmaxf(yhi, yv);
      ylo = fminf(ylo, yv);
    }
  };
  int64_t row = (int64_t)blockIdx.x * RL + rl;
  if (vec) {
#pragma nounroll
    for (; row + 3 * stride < n; row += 4 * stride) {
      f32x4v q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const f32x4v*>(X + (row + u * stride) * D + c0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float v[4] = {q[u][0], q[u][1], q[u][2], q[u][3]};
        take(v, row + u * stride);
      }
    }
  }
#pragma nounroll
  for (; row < n; row += stride) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = c0 + j < dr ? X[row * dr + c0 + j] : 0.f;
    take(v, row);
  }
  __shared__ float smn[256][4], smx[256][4], syhi[256], sylo[256];
  __shared__ double ssum[256][4], sy[256];
#pragma unroll
  for (int j = 0; j < 4; ++j) { smn[threadIdx.x][j] = mn[j]; smx[threadIdx.x][j] = mx[j]; ssum[threadIdx.x][j] = sum[j]; }
  sy[threadIdx.x] = ysum;
  syhi[threadIdx.x] = yhi;
  sylo[threadIdx.x] = ylo;
  __syncthreads();
  if (rl == 0) {
#pragma nounroll
    for (int q = 1; q < RL; ++q) {       // (unrolled, its LDS reads are all hoisted: 256 registers, one wave per SIMD)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mn[j] = fminf(mn[j], smn[q * CGR + cg][j]);
        mx[j] = fmaxf(mx[j], smx[q * CGR + cg][j]);
        sum[j] += ssum[q * CGR + cg][j];
      }
      if (cg == 0) {
        ysum += sy[q * CGR];
        yhi = fmaxf(yhi, syhi[q * CGR]);
        ylo = fminf(ylo, sylo[q * CGR]);
      }
    }
    // pmax[block]: column minima | column maxima | largest y | smallest y
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pmax[(int64_t)blockIdx.x * (2 * D + 2) + c0 + j] = mn[j];
      pmax[(int64_t)blockIdx.x * (2 * D + 2) + D + c0 + j] = mx[j];
      psum[(int64_t)blockIdx.x * (D + 1) + c0 + j] = sum[j];
    }
    if (cg == 0) {
      pmax[(int64_t)blockIdx.x * (2 * D + 2) + 2 * D] = yhi;
      pmax[(int64_t)blockIdx.x * (2 * D + 2) + 2 * D + 1] = ylo;
      psum[(int64_t)blockIdx.x * (D + 1) + D] = ysum;
    }
  }
}

// The header from the partials of f16_colstats_kernel: the shift vector (column means in f32, 0 where not finite),
// the scales and sums of the centred columns, the statistics of y.
template <int D>
__global__ __launch_bounds__(1024) void f16_header_kernel(const float* __restrict__ pmax, const double* __restrict__ psum,
                                                          int nblk, int64_t n, Header* __restrict__ hdr) {
  // 1024 threads = 16 slices x 64 columns: slice s folds the partials s, s + 16, ... of its column, then column
  // threads of slice 0 fold the 16 slices -- the same order every run (a single thread per column walking all
  // 1024 partials took half a millisecond)
  __shared__ float fmn[16][64], fmx[16][64], fhi[16], flo[16];
  __shared__ double fsm[16][64], fys[16];
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  float cmn = INFINITY, cmx = -INFINITY;
  double sum = 0.0;
  if (col < D)
    for (int b = sl; b < nblk; b += 16) {
      cmn = fminf(cmn, pmax[(int64_t)b * (2 * D + 2) + col]);
      cmx = fmaxf(cmx, pmax[(int64_t)b * (2 * D + 2) + D + col]);
      sum += psum[(int64_t)b * (D + 1) + col];
    }
  fmn[sl][col] = cmn;
  fmx[sl][col] = cmx;
  fsm[sl][col] = sum;
  if (col == 0) {
    double ys = 0.0;
    float yhi = -INFINITY, ylo = INFINITY;
    for (int b = sl; b < nblk; b += 16) {
      ys += psum[(int64_t)b * (D + 1) + D];
      yhi = fmaxf(yhi, pmax[(int64_t)b * (2 * D + 2) + 2 * D]);
      ylo = fminf(ylo, pmax[(int64_t)b * (2 * D + 2) + 2 * D + 1]);
    }
    fys[sl] = ys; fhi[sl] = yhi; flo[sl] = ylo;
  }
  __syncthreads();
  if (sl != 0) return;
  for (int q = 1; q < 16; ++q) {
    cmn = fminf(cmn, fmn[q][col]);
    cmx = fmaxf(cmx, fmx[q][col]);
    sum += fsm[q][col];
  }
  if (col < D) {
    const float mean = n > 0 ? (float)(sum / (double)n) : 0.f;
    const float shift = isfinite(mean) ? mean : 0.f;
    // largest |x - shift| of the column, rounded as the plane's values are: at one of its two ends
    const float amax = n > 0 ? fmaxf(fabsf((float)((double)cmx - (double)shift)), fabsf((float)((double)cmn - (double)shift)))
                             : 0.f;
    hdr->shift[col] = shift;
    hdr->colscale[col] = amax > 0.f ? col_scale(amax) : 0.f;   // 0 marks an all-zero (here: constant) column
    hdr->colsum[col] = sum - (double)n * (double)shift;         // sum of the centred column
  } else {
    hdr->shift[col] = 0.f;
    hdr->colscale[col] = 0.f;
    hdr->colsum[col] = 0.0;
  }
  if (col == 0) {
    double ys = 0.0;
    float yhi = -INFINITY, ylo = INFINITY;
    for (int q = 0; q < 16; ++q) {
      ys += fys[q];
      yhi = fmaxf(yhi, fhi[q]);
      ylo = fminf(ylo, flo[q]);
    }
    const float ybar = n > 0 ? (float)(ys / (double)n) : 0.f;
    hdr->ysum = ys;
    hdr->ybar = isfinite(ybar) ? ybar : 0.f;
    hdr->ymax = n > 0 ? fmaxf(fabsf(yhi - hdr->ybar), fabsf(ylo - hdr->ybar)) : 0.f;
  }
}

// ---- pack-time sums (once per dataset) -------------------------------------------------------------------------
// With h the f16 plane (as numbers), x' = (x - shift) s the exact scaled entry and r = x' - h what the plane leaves
// out, the finalize kernel needs, in double,
//   X^T X = H^T H + A,   A = X^T X - H^T H = P + P^T + R^T R,   Bt = P,   P[f][g] = sum_i h_if r_ig,
//   XR^T y = sum_i r_i y_i,   XR^T 1 = sum_i r_i
// (all brought back to the columns' own units at the end: the scales are powers of two).  The three 64 x 64 products
// run on the f16 matrix cores as sums over ONE 64-row tile in f32, which each wave then adds to its running totals
// in double: a tile sum carries ~1e-7 of its own size, independent from tile to tile, and the totals over 10^5 tiles
// end up good to ~1e-9 of the sums of magnitudes -- far inside what the compensation needs (its terms are
// 2^-12-sized corrections of sums that are themselves held to 1e-5).  The two vectors are summed in double by the
// staging threads from the exact r.  Everything is added in a fixed order: the packed dataset is bit-for-bit the
// same from run to run.  (The first version of these sums ran as 4 x 4 register tiles of double-precision fma on the
// vector ALU, bound by its LDS reads: 24 ms of a 30 ms pack step at N = 10M, D = 64; the second on
// v_mfma_f32_16x16x4_f32 over f32 planes.)  D[m][n] of a product: lane l, register i holds m = 4 (l >> 4) + i,
// n = l & 15.
constexpr int kGramBlocks = 512;          // two to three resident per CU: one stages while another multiplies
constexpr double kRScale = 2048.0;         // the remainder planes hold 2^11 r (see f16_pack_sums_kernel)
constexpr int kSumStride = 3 * 64 * 64 + 128;     // per block: H^T H | R^T R | P | XR^T y | XR^T 1 (doubles)
static_assert(kSumStride == 64 * 64 + kLowStride, "scratch layout");

// tile t of a role -> feature tiles (ft, gt): roles 0, 1 walk the upper triangle row by row, roles 2, 3 take the
// upper and lower half of P's tile rows
constexpr int sums_ft(int role, int ntl, int t) {
  if (role >= 2) return (role - 2) * (ntl / 2) + t / ntl;
  int ft = 0, first = 0;
  while (t >= first + (ntl - ft)) { first += ntl - ft; ++ft; }
  return ft;
}
constexpr int sums_gt(int role, int ntl, int t) {
  if (role >= 2) return t % ntl;
  int ft = 0, first = 0;
  while (t >= first + (ntl - ft)) { first += ntl - ft; ++ft; }
  return ft + (t - first);
}
constexpr int sums_tiles(int role, int ntl) { return role < 2 ? ntl * (ntl + 1) / 2 : ntl * ntl / 2; }

template <int D, int ROLE, int NT>
__device__ __forceinline__ void sums_store(double* __restrict__ out, int ln, int g, const double (&accd)[NT][4]) {
  constexpr int NTL = D / 16, NTW = sums_tiles(ROLE, NTL);
  constexpr int base = ROLE == 0 ? 0 : ROLE == 1 ? 4096 : 8192;
  static_for<NTW>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    constexpr int ft = sums_ft(ROLE, NTL, t), gt = sums_gt(ROLE, NTL, t);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[base + (16 * ft + 4 * g + i) * 64 + 16 * gt + ln] = accd[t][i];
  });
}

// The plane and the sums in one pass over X.  Workgroup = 256 threads over 64-row tiles: every thread turns one or two
// 8-column pieces of a row into the plane's f16 values h (written to the packed buffer), the remainder r = x' - h in
// double and the two f16 pieces r1 + r2 of 2^11 r (r to 2^-22 of itself), and
// stores h, r1, r2 as three swizzled f16 images in LDS; the four waves then take one role each -- H^T H (upper
// tiles), R1^T R1 (upper tiles: R^T R to 2^-10 of itself, a term 2^-24 the size of H^T H), and the two halves of
// P = H^T (R1 + R2) -- as v_mfma_f32_16x16x32_f16 products whose operands are both transposed reads of those images
// (ds_read_tr16_b64: lane = feature, eight rows of a 32-row step each, the same row order on both sides).  Every
// product of two f16 numbers is exact in f32; a tile's 64 addends are summed in the f32 accumulator and the tile
// sums added to running totals in double, in a fixed order.  (The previous version multiplied f32 planes with
// v_mfma_f32_16x16x4_f32, a sixteenth of the rate, in a pass of its own: 4.4 + 1.3 ms at N = 10M, D = 64.)
template <int D>
__global__ __launch_bounds__(256) void f16_pack_sums_kernel(const float* __restrict__ X, const float* __restrict__ y,
                                                            int64_t n, int dr, int64_t n_pad,
                                                            unsigned char* __restrict__ packed,
                                                            double* __restrict__ part) {
  constexpr int NTL = D / 16;                    // feature tiles per side
  constexpr int TPR = D / 8;                     // 8-column pieces per row
  constexpr int PER = 64 * TPR / 256;            // pieces per thread and tile
  constexpr int SA = 2 * D;                      // LDS row stride in bytes (16-B chunks XOR-swizzled by row)
  constexpr int NT = sums_tiles(0, NTL) > sums_tiles(2, NTL) ? sums_tiles(0, NTL) : sums_tiles(2, NTL);
  static_assert(PER >= 1 && 256 % TPR == 0, "staging shape");
  __shared__ __attribute__((aligned(16))) unsigned char img[3][64 * SA];   // h | r1 | r2
  __shared__ double colred[2][256][8];
  const Header* hdr = reinterpret_cast<const Header*>(packed);
  unsigned char* planes = packed + kHeader;
  unsigned char* yplane = planes + (int64_t)n_pad * D * 2;
  const int tid = threadIdx.x, lane = tid & 63, ln = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto swz = [](int row) { return D == 64 ? (row & 7) : ((0x78 >> (2 * ((row >> 2) & 3))) & 3); };
  const int scol = (tid % TPR) * 8;              // this thread's columns (the same for all its pieces)
  float shf[8], csf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool cin = scol + j < dr;
    shf[j] = cin ? hdr->shift[scol + j] : 0.f;
    csf[j] = cin ? hdr->colscale[scol + j] : 0.f;
  }
  double vy[8], v1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { vy[j] = 0.0; v1[j] = 0.0; }
  double accd[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) accd[t][i] = 0.0;
  // transposed fragment of feature tile ft, 16-row half: lane (ln, g) -> row 4 g + (ln >> 2) of the half
  int offX[NTL];
#pragma unroll
  for (int ft = 0; ft < NTL; ++ft) {
    const int trow = 4 * g + (ln >> 2);          // swz(trow + 16) == swz(trow)
    offX[ft] = trow * SA + (((2 * ft + ((ln & 3) >> 1)) ^ swz(trow)) << 4) + 8 * (ln & 1);
  }
  auto frag = [&](int plane, int step, int ft) {
    const unsigned char* base = img[plane] + 32 * step * SA + offX[ft];
    const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
    const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 16 * SA));
    return __builtin_bit_cast(f16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  const int64_t ntile = n_pad / 64;
  // raw rows of a tile, one tile ahead: the loads of tile t + 1 are in flight while the waves multiply tile t
  float v[PER][8], yv[PER];
  auto load_tile = [&](int64_t t0) {
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int64_t row = t0 * 64 + (tid + e * 256) / TPR;
      const bool ok = t0 < ntile && row < n;
      if (ok && dr == D) {
        const f32x4v* p = reinterpret_cast<const f32x4v*>(X + row * D + scol);
        const f32x4v a = p[0], b = p[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[e][j] = a[j]; v[e][4 + j] = b[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[e][j] = (ok && scol + j < dr) ? X[row * dr + scol + j] : 0.f;
      }
      yv[e] = ok ? y[row] : 0.f;
    }
  };
  load_tile(blockIdx.x);
  for (int64_t t0 = blockIdx.x; t0 < ntile; t0 += gridDim.x) {
    __syncthreads();                             // the previous tile's fragments have been read
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int idx = tid + e * 256;
      const int srow = idx / TPR, ch = idx % TPR;
      const int64_t row = t0 * 64 + srow;
      const bool ok = row < n;
      f16x8 fh, f1, f2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool in = ok && scol + j < dr;
        // x' = (x - shift) s exactly, in f32 pieces: x - shift = d + err (two-sum), s is a power of two, so d s is
        // the f32 value plane_value() rounds to f16 and the part the plane leaves out is (d s - h) + err s, both
        // terms exact in f32; their f32 sum r is good to 2^-24 of itself.  Padding rows and columns stay zero.
        const float nsh = -shf[j];
        const float dlt = v[e][j] + nsh;
        const float bb = dlt - v[e][j];
        const float err = (v[e][j] - (dlt - bb)) + (nsh - bb);
        const float xs = dlt * csf[j];
        const _Float16 h = in ? (_Float16)xs : (_Float16)0.f;
        const float r = in ? (xs - (float)h) + err * csf[j] : 0.f;
        // the pieces of r are taken from 2^11 r: r is at most half a unit in the last place of h, and unscaled its
        // second piece would fall among f16's subnormals (2^-24 apart) and lose what it is there to keep
        const float rs = r * (float)kRScale;
        const _Float16 r1 = (_Float16)rs;
        fh[j] = h;
        f1[j] = r1;
        f2[j] = (_Float16)(rs - (float)r1);
        vy[j] += (double)(r * yv[e]);
        v1[j] += (double)r;
      }
      *reinterpret_cast<f16x8*>(planes + (row * D + scol) * 2) = fh;
      if (ch == 0) *reinterpret_cast<float*>(yplane + row * 4) = yv[e];
      const int off = srow * SA + ((ch ^ swz(srow)) << 4);
      *reinterpret_cast<f16x8*>(img[0] + off) = fh;
      *reinterpret_cast<f16x8*>(img[1] + off) = f1;
      *reinterpret_cast<f16x8*>(img[2] + off) = f2;
    }
    load_tile(t0 + gridDim.x);
    __syncthreads();
    // one role per wave; the branches are wave-uniform and every wave comes back to the loop's two barriers
    auto role_tile = [&](auto role_c) {
      constexpr int ROLE = decltype(role_c)::value;
      constexpr int NTW = sums_tiles(ROLE, NTL);
      constexpr double unscale = ROLE == 0 ? 1.0 : ROLE == 1 ? 1.0 / (kRScale * kRScale) : 1.0 / kRScale;
      f32x4v accf[NTW];
#pragma unroll
      for (int t = 0; t < NTW; ++t) accf[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int step = 0; step < 2; ++step) {
        f16x8 fa[NTL], fb[NTL], fc[NTL];         // A side; B side (and its second piece for P)
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
          if (ROLE == 0) { fa[t] = frag(0, step, t); fb[t] = fa[t]; }
          else if (ROLE == 1) { fa[t] = frag(1, step, t); fb[t] = fa[t]; }
          else {
            if (t / (NTL / 2) == ROLE - 2) fa[t] = frag(0, step, t);
            fb[t] = frag(1, step, t);
            fc[t] = frag(2, step, t);
          }
        }
        static_for<NTW>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          constexpr int ft = sums_ft(ROLE, NTL, t), gt = sums_gt(ROLE, NTL, t);
          accf[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ft], fb[gt], accf[t], 0, 0, 0);
          if constexpr (ROLE >= 2) accf[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ft], fc[gt], accf[t], 0, 0, 0);
        });
      }
#pragma unroll
      for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) accd[t][i] = fma((double)accf[t][i], unscale, accd[t][i]);
    };
    if (wave == 0) role_tile(std::integral_constant<int, 0>{});
    else if (wave == 1) role_tile(std::integral_constant<int, 1>{});
    else if (wave == 2) role_tile(std::integral_constant<int, 2>{});
    else role_tile(std::integral_constant<int, 3>{});
  }

  double* out = part + (int64_t)blockIdx.x * kSumStride;
  if (wave == 0) sums_store<D, 0, NT>(out, ln, g, accd);
  else if (wave == 1) sums_store<D, 1, NT>(out, ln, g, accd);
  else if (wave == 2) sums_store<D, 2, NT>(out, ln, g, accd);
  else sums_store<D, 3, NT>(out, ln, g, accd);
  // the two vectors: the threads that share a column piece add up in a fixed order
#pragma unroll
  for (int j = 0; j < 8; ++j) { colred[0][tid][j] = vy[j]; colred[1][tid][j] = v1[j]; }
  __syncthreads();
  if (tid < TPR) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      double sy = 0.0, s1 = 0.0;
      for (int q = tid; q < 256; q += TPR) {
        sy += colred[0][q][j];
        s1 += colred[1][q][j];
      }
      out[3 * 4096 + scol + j] = sy;
      out[3 * 4096 + 64 + scol + j] = s1;
    }
  }
}

__device__ __forceinline__ double iq0(const Header* hdr, int q) {
  const float c = hdr->colscale[q];
  return c > 0.f ? 1.0 / (double)c : 0.0;
}

// stage 2: fixed-order sums over the blocks, assembled into X^T X (gram) and A | Bt | XR^T y | XR^T 1 (low) in the
// columns' own units; entries beyond the model's features are zero.  Eight lanes per matrix entry (p, q): lane s adds
// blocks s, s + 8, ... in order, then the eight are added in a fixed butterfly.
constexpr int kFinalSplit = 8;
__device__ __forceinline__ double split_sum(double v) {
#pragma unroll
  for (int m = 1; m < kFinalSplit; m <<= 1) v += __shfl_xor(v, m);
  return v;
}
__global__ __launch_bounds__(256) void f16_sums_final(const double* __restrict__ part, int nblk, int D,
                                                      const Header* __restrict__ hdr, double* __restrict__ gram,
                                                      double* __restrict__ low) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = t / kFinalSplit, sub = t % kFinalSplit;      // the grid covers 64 * 64 entries exactly
  const int p = e >> 6, q = e & 63;
  const bool in = p < D && q < D;
  double hh = 0.0, rr = 0.0, pq = 0.0, qp = 0.0;
  if (in) {
    // the symmetric products hold their upper tiles only
    const int su = (p >> 4) <= (q >> 4) ? p * 64 + q : q * 64 + p;
    for (int b = sub; b < nblk; b += kFinalSplit) {
      const double* o = part + (int64_t)b * kSumStride;
      hh += o[su];
      rr += o[4096 + su];
      pq += o[8192 + p * 64 + q];
      qp += o[8192 + q * 64 + p];
    }
  }
  hh = split_sum(hh); rr = split_sum(rr); pq = split_sum(pq); qp = split_sum(qp);
  double s = 0.0;                                  // rows 0, 1 of the grid also carry the two vectors
  if (p < 2 && q < D)
    for (int b = sub; b < nblk; b += kFinalSplit) s += part[(int64_t)b * kSumStride + 3 * 4096 + 64 * p + q];
  s = split_sum(s);
  if (sub != 0) return;
  const float cp = in ? hdr->colscale[p] : 0.f, cq = in ? hdr->colscale[q] : 0.f;
  const double ip = cp > 0.f ? 1.0 / (double)cp : 0.0, iq = cq > 0.f ? 1.0 / (double)cq : 0.0;
  const double a = (pq + qp) + rr;
  gram[e] = (hh + a) * ip * iq;
  low[e] = a * ip * iq;
  low[4096 + e] = pq * ip * iq;
  if (p < 2) low[8192 + 64 * p + q] = s * (q < D ? iq0(hdr, q) : 0.0);
}

// finalize, one workgroup of kFinGroups waves per chain: fixed-order reduction over row blocks in double (a wave
// takes every kFinGroups-th row block; consecutive lanes read consecutive sums), undoing the power-of-two scales, the
// closed-form correction for the part of W that its two f16 pieces do not hold (see the top of this file), S1 from
// the packed data sums.  16 waves when there are few chains (the chip is filled by waves per chain), 4 when there
// are many (all workgroups resident at once).
template <int kFinGroups>
__global__ __launch_bounds__(64 * kFinGroups) void linreg_f16_finalize(
    const float* __restrict__ partials, int nb, int64_t n, int d, int dplane, int cpb,
    const unsigned char* __restrict__ packed, const float* __restrict__ theta, int C, float* __restrict__ logp,
    float* __restrict__ grad, const double* __restrict__ ref) {
  const int c = blockIdx.x, q2 = d + 2;
  const int64_t tot = (int64_t)C * q2;
  const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
  __shared__ double part[5][kFinGroups][64];
  __shared__ double sums[66], dw[64], wv[64];
  __shared__ float sh_t, sh_b0c, sh_pad;
  __shared__ double sh_bt, sh_s1;
  const Header* hdr = reinterpret_cast<const Header*>(packed);
  const double* gram = reinterpret_cast<const double*>(packed + kGramOffset);
  const double* low = reinterpret_cast<const double*>(packed + kLowOffset);
  const float* th = theta + (int64_t)c * q2;
  // how the streaming kernel scaled this chain's block: > 0 = one scale for the whole block, targets entering
  // through the accumulator and no intercept at all in the kernel; 0 = one scale and one intercept per chain
  const float block_scale = partials[(int64_t)nb * tot + c / cpb];
  const double nn = (double)n;
  if (grp == 0) {
    // dw_f = w_f - (what the kernel's f16 piece of the scaled weight stands for), recomputed with the kernel's
    // own arithmetic: w'' = -w / s_f * t, piece w1, remainder exact in f32
    const float cs = o < d ? hdr->colscale[o] : 0.f;
    const float v = o < d ? -scaled_weight(th[1 + o], cs) : 0.f;
    float amax = fabsf(v);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    const float b0cen = centred_intercept_wave(th, hdr->shift, d, o, hdr->ybar);
    const float t = block_scale > 0.f ? block_scale : chain_scale(amax, b0cen, hdr->ymax, dplane);
    float grid, ginv;
    piece_grid(amax * t, grid, ginv);
    const float q = v * t;
    const float rem = q - (float)first_piece(q, grid, ginv);        // exact: the piece is q rounded to a coarser grid
    dw[o] = cs > 0.f ? -(double)rem * (double)cs / (double)t : 0.0;
    wv[o] = o < d ? (double)th[1 + o] : 0.0;
    // the exact intercept in the centred frame, b0 + shift . w, and S1 = sum_i r_i from the packed sums
    double mw = o < d ? (double)hdr->shift[o] * (double)th[1 + o] : 0.0;
    double wc = o < d ? (double)th[1 + o] * hdr->colsum[o] : 0.0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      mw += __shfl_xor(mw, off, 64);
      wc += __shfl_xor(wc, off, 64);
    }
    const float padv = pad_value(centred_intercept_wave(theta + (int64_t)(c / cpb) * cpb * q2, hdr->shift, d, o, hdr->ybar));
    if (o == 0) {
      sh_t = t;
      sh_pad = padv;
      sh_b0c = block_scale > 0.f ? 0.f : b0cen;      // the intercept (relative to ybar) the kernel worked with
      sh_bt = (double)th[0] + mw;
      sh_s1 = hdr->ysum - nn * ((double)th[0] + mw) - wc;
    }
  }
  {
    // the chain's q2 = d + 2 <= 66 raw sums over the row blocks: lane o takes sum o and, for the one or two sums
    // beyond the 64th, a second accumulator in the same loop (one pass over the partials, one exchange)
    const float* pc = partials + (int64_t)c * q2;
    const bool first = o < q2, extra = 64 + o < q2;
    double s = 0.0, sx = 0.0;
    // (loads unconditional, from clamped indices, so that a whole unrolled batch is in flight at once)
    const int i0 = first ? o : 0, i1 = extra ? 64 + o : i0;
    if (q2 > 64) {
#pragma unroll 8
      for (int b = grp; b < nb; b += kFinGroups) {
        s += (double)pc[(int64_t)b * tot + i0];
        sx += (double)pc[(int64_t)b * tot + i1];
      }
    } else {
#pragma unroll 16
      for (int b = grp; b < nb; b += kFinGroups) s += (double)pc[(int64_t)b * tot + i0];
    }
    part[0][grp][o] = s;
    part[1][grp][o] = sx;
    __syncthreads();
    if (grp == 0) {
#pragma unroll 4
      for (int w = 1; w < kFinGroups; ++w) {      // (not unrolled all the way: 16 waves' worth of doubles in flight spill)
        s += part[0][w][o];
        sx += part[1][w][o];
      }
      if (first) sums[o] = s;
      if (extra) sums[64 + o] = sx;
    }
    __syncthreads();
  }
  const double u = (double)sh_t;       // the chain's scale: S2' = t^2 S2_k, G'_f = t s_f G_k,f
  // the kernel's intercept is ybar + (the float it worked with: the centred intercept summed in float, or none at
  // all); what that leaves of the exact centred intercept bt = b0 + shift . w is handled like dw, in closed form
  const float b0c = sh_b0c;
  const double bt = sh_bt;
  const double b0_eff = (double)hdr->ybar + (double)b0c;
  const double db = bt - b0_eff;
  const double sc = (double)th[d + 1];
  const double inv2 = 1.0 / (sc * sc);
  {
    // (X^T X dw)_f, (A w)_f, (A dw)_f (both matrices symmetric), (B w)_f, (B dw)_f: a wave takes every
    // kFinGroups-th g
    double cf = 0.0, caw = 0.0, cad = 0.0, cbw = 0.0, cbd = 0.0;
    if (o < d)
      for (int g = grp; g < d; g += kFinGroups) {
        const double am = low[g * 64 + o], bm = low[4096 + g * 64 + o];
        cf = fma(gram[g * 64 + o], dw[g], cf);
        caw = fma(am, wv[g], caw);
        cad = fma(am, dw[g], cad);
        cbw = fma(bm, wv[g], cbw);
        cbd = fma(bm, dw[g], cbd);
      }
    part[0][grp][o] = cf;
    part[1][grp][o] = caw;
    part[2][grp][o] = cad;
    part[3][grp][o] = cbw;
    part[4][grp][o] = cbd;
  }
  __syncthreads();
  if (grp == 0) {
    const int f = o;
    double sv = 0.0;      // this feature's share of the corrections to S2
    if (f < d) {
      const double cs = (double)hdr->colscale[f];
      double m[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) m[k] = part[k][0][f];
#pragma unroll 2
      for (int w = 1; w < kFinGroups; ++w) {      // (two waves' five sums at a time: all 80 at once spill)
#pragma unroll
        for (int k = 0; k < 5; ++k) m[k] += part[k][w][f];
      }
      // kernel: r_k = y - b0_eff - xh.wk with wk = w - dw and b0_eff = b0 - db, G_k = sum xh r_k
      // (G'_f = t s_f G_k,f).  The true residual is r = r_k - (db + xr.w + xh.dw), hence
      //   G = G_k - A wk - (X^T X) dw + t1 - db XH^T 1,   t1 = XR^T y - b0 XR^T 1,   XH^T 1 = X^T 1 - XR^T 1
      //   S2 = S2_k - 2 (w . cr + dw . G_k) + (w . A w - 2 w . B wk + dw . (X^T X - A) dw)
      //        - 2 db (sum y - n b0_eff - w . X^T 1) + n db^2,   cr = t1 + db XR^T 1 - B wk = sum xr r_k
      // (db: -2 db sum r_k from the cross term, + 2 db (w . XR^T 1 + dw . XH^T 1) from the square of what was left
      // out)
      const double g0 = cs > 0.0 ? sums[2 + f] / (u * cs) : sums[2 + f] * 0.0;
      const double xr1 = low[8192 + 64 + f];
      const double sxh = hdr->colsum[f] - xr1;
      const double t1 = low[8192 + f] - bt * xr1;
      const double bwk = m[3] - m[4];
      sv = -2.0 * (wv[f] * (t1 + db * xr1 - bwk) + dw[f] * g0) + wv[f] * (m[1] - 2.0 * bwk) + dw[f] * (m[0] - m[2]) +
           2.0 * db * wv[f] * (sxh + xr1);
      // (the sums are those of the centred columns: d/dw_f = sum_i (x~_if + shift_f) r_i)
      if (grad)
        grad[(int64_t)c * q2 + 1 + f] =
            (float)((g0 - (m[1] - m[2]) - m[0] + t1 - db * sxh + (double)hdr->shift[f] * sh_s1) * inv2);
    }
    // sum over the features: a butterfly (fixed order), every lane ends with the total
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sv += __shfl_xor(sv, off, 64);
    if (o == 0) {
      // padding rows of the last tile came through as r' = fma(ypad, t, -t (b0 - ybar)): out again
      // (with one scale per block the padding rows are exact zeros)
      const double npad = (double)((n + kTileRows - 1) / kTileRows * kTileRows - n);
      const float rpad = block_scale > 0.f ? 0.f : fmaf(sh_pad, sh_t, -b0c * sh_t);
      double s2 = (sums[0] + sums[1] - npad * (double)rpad * (double)rpad) / (u * u);   // S2' arrives as a float pair
      s2 += -2.0 * db * (hdr->ysum - nn * b0_eff) + nn * db * db;
      s2 += sv;
      logp[c] = (float)(-0.5 * s2 * inv2 - nn * log(sc) - 0.5 * nn * BVI_LN2PI_D - (ref ? *ref : 0.0));
      if (grad) grad[(int64_t)c * q2 + d + 1] = (float)(-nn / sc + s2 * inv2 / sc);
    } else if (o == 1 && grad) {
      grad[(int64_t)c * q2] = (float)(sh_s1 * inv2);
    }
  }
}

int64_t pad_rows(int64_t n) { return (n + kTileRows - 1) / kTileRows * kTileRows; }
constexpr int kStatBlocks = 1024;

}  // namespace

// plane columns for a model with d features
static int plane_cols(int d) { return d <= 32 ? 32 : 64; }

size_t bvi_linreg_f16_packed_size(int64_t n, int d) {
  const int64_t np = pad_rows(n);
  return (size_t)(kHeader + np * plane_cols(d) * 2 + np * 4);
}

// scratch for the column statistics of the pack step (bytes)
size_t bvi_linreg_f16_pack_scratch(int d) {
  const int dp = plane_cols(d);
  return (size_t)kStatBlocks * (2 * dp + 2) * sizeof(float) + (size_t)kStatBlocks * (dp + 1) * sizeof(double) +
         (size_t)kGramBlocks * kSumStride * sizeof(double) + 16;
}

int bvi_launch_linreg_f16_pack(const float* X, const float* y, int64_t n, int d, void* packed, void* scratch,
                               hipStream_t s) {
  if (((uintptr_t)X & 15) != 0 || ((uintptr_t)packed & 15) != 0 || ((uintptr_t)scratch & 7) != 0)
    return BVI_ERR_ARG;
  if (d < 1 || d > 64) return BVI_ERR_UNSUPPORTED;
  const int dp = plane_cols(d);
  const int64_t np = pad_rows(n);
  double* psum = (double*)scratch;
  double* gpart = psum + (size_t)kStatBlocks * (dp + 1);
  float* pmax = (float*)(gpart + (size_t)kGramBlocks * kSumStride);
  double* gram = (double*)((unsigned char*)packed + kGramOffset);
  double* low = (double*)((unsigned char*)packed + kLowOffset);
  if (dp == 64) {
    hipLaunchKernelGGL(f16_colstats_kernel<64>, dim3(kStatBlocks), dim3(256), 0, s, X, y, n, d, pmax, psum);
    hipLaunchKernelGGL(f16_header_kernel<64>, dim3(1), dim3(1024), 0, s, (const float*)pmax, (const double*)psum,
                       kStatBlocks, n, (Header*)packed);
    hipLaunchKernelGGL(f16_pack_sums_kernel<64>, dim3(kGramBlocks), dim3(256), 0, s, X, y, n, d, np, (unsigned char*)packed,
                       gpart);
  } else {
    hipLaunchKernelGGL(f16_colstats_kernel<32>, dim3(kStatBlocks), dim3(256), 0, s, X, y, n, d, pmax, psum);
    hipLaunchKernelGGL(f16_header_kernel<32>, dim3(1), dim3(1024), 0, s, (const float*)pmax, (const double*)psum,
                       kStatBlocks, n, (Header*)packed);
    hipLaunchKernelGGL(f16_pack_sums_kernel<32>, dim3(kGramBlocks), dim3(256), 0, s, X, y, n, d, np, (unsigned char*)packed,
                       gpart);
  }
  hipLaunchKernelGGL(f16_sums_final, dim3(64 * 64 * kFinalSplit / 256), dim3(256), 0, s, (const double*)gpart, kGramBlocks, dp,
                     (const Header*)packed, gram, low);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

template <int D, int NW, int CT>
static int launch_f16(const unsigned char* pk, int64_t n, int d, const float* theta, int c, float* logp, float* grad,
                      float* ws, hipStream_t s, const double* logp_ref) {
  constexpr int cpb = 16 * CT * NW;
  // NW = 8: one workgroup per CU; NW = 4 (at most 128 chains): two per CU, each on its own row blocks
  // (one tile per wave: three per CU on a large dataset -- the kernel is then bound by its own HBM stream, and
  // a third workgroup's loads in flight are worth 3 - 5 %; a mid-size dataset has too few tiles for that)
  const int max_blocks = NW == 8 ? kMaxRowBlocks : ((CT == 1 && n >= 1000000) ? 3 : 2) * kMaxRowBlocks;
  const int ncb = bvi_ceil_div(c, cpb);
  int per8 = (max_blocks / 8) / ncb;
  // (no more row blocks than the dataset has tiles: a small dataset would launch workgroups with nothing to do and
  // the finalize kernel would add their zeros)
  const int64_t tiles8 = ((n + kTileRows - 1) / kTileRows + 7) / 8;
  if (per8 > tiles8) per8 = (int)tiles8;
  if (per8 < 1) per8 = 1;
  const int nrb = 8 * per8;
  // workspace: partials [nrb][c][d+2]
  constexpr bool pipelined = CT == 2;      // many chains: the software-pipelined form of the pass, 3 staged tiles
  const size_t lds = (pipelined ? 3 : 2) * (size_t)(kTileRows * 2 * D + kTileRows * 4);
  // > 64 KB of dynamic LDS has to be allowed once per kernel and device (setting it twice from two threads is
  // harmless)
  static std::atomic<bool> lds_ok[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return BVI_ERR_ARG;
  if (!lds_ok[dev]) {
    hipError_t e = pipelined ? hipFuncSetAttribute((const void*)&linreg_f16_pipe_kernel<D, NW, true>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                             : hipFuncSetAttribute((const void*)&linreg_f16_kernel<D, NW, CT>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && pipelined)
      e = hipFuncSetAttribute((const void*)&linreg_f16_pipe_kernel<D, NW, false>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    lds_ok[dev] = true;
  }
  if constexpr (pipelined) {
    if (grad)
      hipLaunchKernelGGL((linreg_f16_pipe_kernel<D, NW, true>), dim3(ncb * nrb), dim3(64 * NW), lds, s, pk, n, d, theta,
                         c, ncb, nrb, ws);
    else
      hipLaunchKernelGGL((linreg_f16_pipe_kernel<D, NW, false>), dim3(ncb * nrb), dim3(64 * NW), lds, s, pk, n, d,
                         theta, c, ncb, nrb, ws);
  } else
    hipLaunchKernelGGL((linreg_f16_kernel<D, NW, CT>), dim3(ncb * nrb), dim3(64 * NW), lds, s, pk, n, d, theta, c, ncb,
                       nrb, ws);
  BVI_HIP_CHECK_LAUNCH();
  if (c > 256)
    hipLaunchKernelGGL(linreg_f16_finalize<4>, dim3(c), dim3(256), 0, s, (const float*)ws, nrb, n, d, D, cpb, pk, theta,
                       c, logp, grad, logp_ref);
  else
    hipLaunchKernelGGL(linreg_f16_finalize<16>, dim3(c), dim3(1024), 0, s, (const float*)ws, nrb, n, d, D, cpb, pk, theta,
                       c, logp, grad, logp_ref);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_launch_linreg_f16(const void* packed, int64_t n, int d, const float* theta, int c, float* logp,
                          float* grad, float* ws, hipStream_t s, const double* logp_ref) {
  if (((uintptr_t)packed & 15) != 0) return BVI_ERR_ARG;
  if (d < 1 || d > 64) return BVI_ERR_UNSUPPORTED;
  const unsigned char* pk = (const unsigned char*)packed;
  // Above 128 chains the 8-wave workgroup wins: every tile is fetched from L2 by half as many workgroups
  // (measured at 1024 chains: 6.5 ms against 7.1 ms).  Up to 128 chains 4-wave workgroups, two per CU; up to 64
  // chains with one 16-chain tile per wave, which halves the matrix work per tile and leaves the kernel close
  // to the memory system's pace (N = 10M, D = 64, 64 chains: 0.73 ms against 1.05 ms).
  const bool wide = plane_cols(d) == 64;
  if (c <= 64)
    return wide ? launch_f16<64, 4, 1>(pk, n, d, theta, c, logp, grad, ws, s, logp_ref)
                : launch_f16<32, 4, 1>(pk, n, d, theta, c, logp, grad, ws, s, logp_ref);
  if (c <= 128)
    return wide ? launch_f16<64, 4, 2>(pk, n, d, theta, c, logp, grad, ws, s, logp_ref)
                : launch_f16<32, 4, 2>(pk, n, d, theta, c, logp, grad, ws, s, logp_ref);
  return wide ? launch_f16<64, 8, 2>(pk, n, d, theta, c, logp, grad, ws, s, logp_ref)
              : launch_f16<32, 8, 2>(pk, n, d, theta, c, logp, grad, ws, s, logp_ref);
}
