// Shared device helpers and launch policy for libbvi_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bvi.h"

#define BVI_LN2PI 1.8378770664093453f
#define BVI_LN2PI_D 1.8378770664093453
#define BVI_LOG2E 1.4426950408889634f
#define BVI_LN2 0.6931471805599453f

// Row-block count of the streaming kernels: enough workgroups to fill the
// 256 CUs several times over, capped so the per-block partial sums stay small.
#define BVI_MAX_ROW_BLOCKS 1024

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

static inline int bvi_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Smallest supported lanes-per-wave chain packing >= c (1, 4, 16 or 64).
static inline int bvi_chain_pack(int c) { return c <= 1 ? 1 : c <= 4 ? 4 : c <= 16 ? 16 : 64; }

#define BVI_HIP_CHECK_LAUNCH()                   \
  do {                                           \
    hipError_t e__ = hipGetLastError();          \
    if (e__ != hipSuccess) return (int)e__;      \
  } while (0)

// Per-family launchers (defined in the .hip files), all: partial sums into ws,
// then a finalize kernel that reduces them in fixed order.
// logp_ref (device pointer or null): a level subtracted from every log-likelihood in double before it is rounded
// to float, so that callers who only need differences keep them exact when |logp| is large.
int bvi_launch_normal_iid(const float* y, int64_t n, const float* theta, int c, float* logp, float* grad,
                          float* ws, hipStream_t s, const double* logp_ref = nullptr);
size_t bvi_ws_normal_iid(int64_t n, int c);

int bvi_launch_gmm1d(const float* y, int64_t n, int k, const float* theta, int c, float* logp, float* grad,
                     float* ws, hipStream_t s, const double* logp_ref = nullptr);
size_t bvi_ws_gmm1d(int64_t n, int c, int k);

int bvi_launch_linreg(const float* X, const float* y, int64_t n, int d, const float* theta, int c, int prec,
                      float* logp, float* grad, float* ws, hipStream_t s, const double* logp_ref = nullptr);
size_t bvi_ws_linreg(int64_t n, int c, int d);
size_t bvi_linreg_packed_size(int64_t n, int d);
int bvi_launch_linreg_pack(const float* X, const float* y, int64_t n, int d, void* packed, hipStream_t s);
int bvi_launch_linreg_packed(const void* packed, int64_t n, int d, const float* theta, int c, int prec,
                             float* logp, float* grad, float* ws, hipStream_t s, const double* logp_ref = nullptr);

int bvi_launch_gmm_fullcov(const float* Y, int64_t n, int d, int k, const float* theta, int c, float* logp,
                           float* grad, float* ws, hipStream_t s, const double* logp_ref = nullptr,
                           const void* packed = nullptr);
size_t bvi_gmm_fullcov_packed_size(int64_t n, int d);
size_t bvi_gmm_fullcov_pack_scratch(int d);
int bvi_launch_gmm_fullcov_pack(const float* Y, int64_t n, int d, void* packed, void* scratch, hipStream_t s);
size_t bvi_ws_gmm_fullcov(int64_t n, int c, int d, int k);
int bvi_launch_mvn_tril(const float* Y, int64_t n, int d, int mode, const float* theta, int c, float* logp,
                        float* grad, float* ws, hipStream_t s, const double* logp_ref = nullptr,
                        const void* packed = nullptr);
size_t bvi_ws_mvn_tril(int64_t n, int c, int d);

size_t bvi_ws_cluster_rows(int64_t n, int d, int m);
int bvi_launch_cluster_rows(const float* Y, int64_t n, int d, int m, float* Yout, int64_t* order, void* ws,
                            hipStream_t s);

size_t bvi_linreg_f16_packed_size(int64_t n, int d);
size_t bvi_linreg_f16_pack_scratch(int d);
int bvi_launch_linreg_f16_pack(const float* X, const float* y, int64_t n, int d, void* packed, void* scratch,
                                 hipStream_t s);
int bvi_launch_linreg_f16(const void* packed, int64_t n, int d, const float* theta, int c, float* logp,
                            float* grad, float* ws, hipStream_t s, const double* logp_ref = nullptr);

#ifdef __HIPCC__
// Sum of partials p[b * stride], b = 0 .. nb-1, in double and in a fixed order.  WAVE = false: the calling
// thread adds them one by one; WAVE = true: the 64 lanes of the calling wave stride over b and combine by
// a butterfly (every lane returns the total) -- for the many-row-block launches of the streaming kernels,
// where a single thread would add up to 2048 partials serially.
template <bool WAVE>
__device__ __forceinline__ double bvi_sum_partials(const float* __restrict__ p, int nb, int64_t stride) {
  double s = 0.0;
  if (!WAVE) {
    for (int b = 0; b < nb; ++b) s += (double)p[(int64_t)b * stride];
    return s;
  }
  for (int b = threadIdx.x & 63; b < nb; b += 64) s += (double)p[(int64_t)b * stride];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  return s;
}

// Issue priority of the calling wave: 1 if `high` (the same in every lane; all lanes active), else 0.  One asm
// statement with the branch inside it, so that the surrounding loop body stays one basic block: a C-level if / else
// splits it, and the compiler then moves instructions across the split that the hand-made schedule had placed (the
// sums of squares of a tile's first half were sunk into its second half).  The flag is compared in a vector
// register: under register pressure a wave-uniform value may live in one, whatever the operand constraint says.
__device__ __forceinline__ void bvi_set_priority(bool high) {
  asm volatile(
      "v_cmp_ne_u32_e32 vcc, 0, %0\n\t"
      "s_nop 4\n\t"
      "s_cbranch_vccz .Lbvi_prio_lo_%=\n\t"
      "s_setprio 1\n\t"
      "s_branch .Lbvi_prio_end_%=\n"
      ".Lbvi_prio_lo_%=:\n\t"
      "s_setprio 0\n"
      ".Lbvi_prio_end_%=:"
      :
      : "v"((int)high)
      : "vcc");
}

// Sum over the lanes that share (lane % CP): butterfly over the upper lane bits.
template <int CP>
__device__ __forceinline__ float bvi_reduce_over_rowsubs(float v) {
#pragma unroll
  for (int off = 32; off >= CP; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
#endif
