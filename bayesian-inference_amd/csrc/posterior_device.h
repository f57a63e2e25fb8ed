// Device-side pieces shared by the posterior-assembly kernels (posterior.hip) and the fused HMC
// kernel (hmc_fused.hip): the block-wise constraining bijector, built-in priors, the chain rule to
// unconstrained space, and counter-based Philox4x32-10 draws.
// Reference semantics: tfp.mcmc.TransformedTransitionKernel around the target closure
// (bayes_vi/inference/mcmc/mcmc.py:153-156) with the model's default bijectors
// (bayes_vi/model/model.py:51-55).
#pragma once
#include "bvi_common.h"

namespace {

struct Blocks {
  int n;
  bvi_param_block b[BVI_MAX_PARAM_BLOCKS];
};

__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__host__ __device__ __forceinline__ int usize_of(const bvi_param_block& b) {
  if (b.bijector == BVI_BIJ_FILL_SCALE_TRIL) return b.size / (b.dim * b.dim) * (b.dim * (b.dim + 1) / 2);
  return b.bijector == BVI_BIJ_SOFTMAX_CENTERED ? b.size - 1 : b.size;
}

// tfp.math.fill_triangular: entry (i, j), j <= i, of the d x d matrix is element
// reshape(concat(x[d:], reverse(x)), [d, d])[i][j] of the packed vector x[d(d+1)/2]
__device__ __forceinline__ int fill_triangular_src(int d, int f) {
  const int m = d * (d + 1) / 2;
  return f < m - d ? d + f : 2 * m - d - 1 - f;
}

// Sum / max over the threads of a workgroup of 64 * nw threads (nw <= 16); every thread calls it and gets
// the result.  `red` is a __shared__ float[16] of the calling kernel.
template <bool MAX>
__device__ __forceinline__ float block_reduce(float v, float* red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float o = __shfl_xor(v, off, 64);
    v = MAX ? fmaxf(v, o) : v + o;
  }
  const int nw = blockDim.x >> 6;
  if (nw <= 1) return v;
  __syncthreads();                 // the previous reduction's readers are done with `red`
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < nw; ++w) r = MAX ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

// threads per chain for the kernels that give one workgroup to each chain's parameter vector
inline int chain_threads(int n_elements) { return n_elements <= 128 ? 64 : (n_elements <= 2048 ? 256 : 1024); }

// theta = f(z) for one chain; returns fldj(z)
__device__ float constrain(const Blocks& bl, const float* __restrict__ z, float* __restrict__ theta) {
  float fldj = 0.f;
  int oz = 0, ot = 0;
  for (int i = 0; i < bl.n; ++i) {
    const bvi_param_block b = bl.b[i];
    if (b.bijector == BVI_BIJ_SOFTMAX_CENTERED) {
      float m = 0.f;  // the appended logit is 0
      for (int j = 0; j < b.size - 1; ++j) m = fmaxf(m, z[oz + j]);
      float s = expf(-m);
      for (int j = 0; j < b.size - 1; ++j) s += expf(z[oz + j] - m);
      const float lse = m + logf(s);
      float sumlog = 0.f;
      for (int j = 0; j < b.size; ++j) {
        const float ly = (j < b.size - 1 ? z[oz + j] : 0.f) - lse;
        theta[ot + j] = expf(ly);
        sumlog += ly;
      }
      fldj += 0.5f * logf((float)b.size) + sumlog;
    } else {
      for (int j = 0; j < b.size; ++j) {
        const float x = z[oz + j];
        if (b.bijector == BVI_BIJ_SOFTPLUS) {
          theta[ot + j] = softplusf(x);
          fldj += -softplusf(-x);
        } else if (b.bijector == BVI_BIJ_EXP) {
          theta[ot + j] = expf(x);
          fldj += x;
        } else {
          theta[ot + j] = x;
        }
      }
    }
    oz += usize_of(b);
    ot += b.size;
  }
  return fldj;
}

// log prior(theta) for one chain; adds d log prior / d theta into gth (may be null)
__device__ float log_prior(const Blocks& bl, const float* __restrict__ theta, float* __restrict__ gth) {
  float lp = 0.f;
  int ot = 0;
  for (int i = 0; i < bl.n; ++i) {
    const bvi_param_block b = bl.b[i];
    if (b.prior == BVI_PRIOR_NORMAL) {
      for (int j = 0; j < b.size; ++j) {
        const float u = (theta[ot + j] - b.a) / b.b;
        lp += -0.5f * u * u - 0.5f * BVI_LN2PI - logf(b.b);
        if (gth) gth[ot + j] += -u / b.b;
      }
    } else if (b.prior == BVI_PRIOR_HALFNORMAL) {
      for (int j = 0; j < b.size; ++j) {
        const float x = theta[ot + j], u = x / b.a;
        lp += x < 0.f ? -INFINITY : (0.5f * logf(2.f / 3.14159265358979f) - logf(b.a) - 0.5f * u * u);
        if (gth) gth[ot + j] += -u / b.a;
      }
    } else if (b.prior == BVI_PRIOR_DIRICHLET) {
      lp += lgammaf(b.a * b.size) - b.size * lgammaf(b.a);
      for (int j = 0; j < b.size; ++j) {
        lp += (b.a - 1.f) * logf(theta[ot + j]);
        if (gth) gth[ot + j] += (b.a - 1.f) / theta[ot + j];
      }
    }
    ot += b.size;
  }
  return lp;
}

// g_z = J(z)^T g_theta + d fldj / dz for one chain
__device__ void chain_rule(const Blocks& bl, const float* __restrict__ z, const float* __restrict__ theta,
                           const float* __restrict__ gth, float* __restrict__ gz) {
  int oz = 0, ot = 0;
  for (int i = 0; i < bl.n; ++i) {
    const bvi_param_block b = bl.b[i];
    if (b.bijector == BVI_BIJ_SOFTMAX_CENTERED) {
      float inner = 0.f;
      for (int j = 0; j < b.size; ++j) inner += theta[ot + j] * gth[ot + j];
      for (int j = 0; j < b.size - 1; ++j)
        gz[oz + j] = theta[ot + j] * (gth[ot + j] - inner) + 1.f - (float)b.size * theta[ot + j];
    } else {
      for (int j = 0; j < b.size; ++j) {
        const float x = z[oz + j];
        if (b.bijector == BVI_BIJ_SOFTPLUS) gz[oz + j] = gth[ot + j] * sigmoidf(x) + sigmoidf(-x);
        else if (b.bijector == BVI_BIJ_EXP) gz[oz + j] = gth[ot + j] * expf(x) + 1.f;
        else gz[oz + j] = gth[ot + j];
      }
    }
    oz += usize_of(b);
    ot += b.size;
  }
}

// ---- Philox4x32-10 (Salmon et al., SC'11) ----
struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// 4 standard normals from one Philox block (two Box-Muller pairs)
__device__ __forceinline__ void normal4(U4 r, float* out) {
  const float r0 = sqrtf(-2.f * logf(u01(r.x))), t0 = 6.283185307179586f * u01(r.y);
  const float r1 = sqrtf(-2.f * logf(u01(r.z))), t1 = 6.283185307179586f * u01(r.w);
  out[0] = r0 * cosf(t0);
  out[1] = r0 * sinf(t0);
  out[2] = r1 * cosf(t1);
  out[3] = r1 * sinf(t1);
}

// counter layout: (step lo, step hi, chain, element block | stream << 28)
__device__ __forceinline__ U4 draw(uint64_t seed, uint64_t step, uint32_t chain, uint32_t blk, uint32_t stream) {
  return philox4x32_10(U4{(uint32_t)step, (uint32_t)(step >> 32), chain, blk | (stream << 28)}, (uint32_t)seed,
                       (uint32_t)(seed >> 32));
}

// host: validate and copy the caller's block list; P / Pu = constrained / unconstrained sizes
// (allow_tril: FILL_SCALE_TRIL blocks are implemented by the wave-per-chain kernels only)
inline int make_blocks(const bvi_param_block* blocks, int n_blocks, Blocks* out, int* P, int* Pu,
                       bool allow_tril = false) {
  if (!blocks || n_blocks < 1 || n_blocks > BVI_MAX_PARAM_BLOCKS) return BVI_ERR_ARG;
  out->n = n_blocks;
  *P = 0;
  *Pu = 0;
  for (int i = 0; i < n_blocks; ++i) {
    const bvi_param_block b = blocks[i];
    if (b.size < 1 || b.bijector < BVI_BIJ_IDENTITY || b.bijector > BVI_BIJ_FILL_SCALE_TRIL ||
        b.prior < BVI_PRIOR_NONE || b.prior > BVI_PRIOR_WISHART_TRIL)
      return BVI_ERR_ARG;
    if (b.prior == BVI_PRIOR_WISHART_TRIL && (b.bijector != BVI_BIJ_FILL_SCALE_TRIL || !(b.a > (float)(b.dim - 1)) || !(b.b > 0.f)))
      return BVI_ERR_ARG;
    if (b.bijector == BVI_BIJ_SOFTMAX_CENTERED && b.size < 2) return BVI_ERR_ARG;
    if (b.bijector == BVI_BIJ_FILL_SCALE_TRIL) {
      if (!allow_tril) return BVI_ERR_UNSUPPORTED;
      if (b.dim < 1 || b.dim > 1024 || b.size % (b.dim * b.dim) != 0 || b.prior == BVI_PRIOR_DIRICHLET)
        return BVI_ERR_ARG;
    }
    out->b[i] = b;
    *P += b.size;
    *Pu += usize_of(b);
  }
  return BVI_OK;
}

}  // namespace
