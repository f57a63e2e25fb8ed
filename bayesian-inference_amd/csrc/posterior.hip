// Device-side posterior assembly and the HMC / leapfrog state updates, so that a
// whole transition runs on the stream without host round trips:
//   log pi(z) = loglik(f(z)) + sum_blocks log prior(f(z)) + fldj(z),   theta = f(z)
// with f the block-wise constraining bijector (forward = constrain).
//
// In the reference this is what tfp.mcmc.TransformedTransitionKernel wraps around
// the target closure (bayes_vi/inference/mcmc/mcmc.py:153-156) and what
// tfp.mcmc.HamiltonianMonteCarlo's leapfrog does with its gradient
// (transition_kernels.py:122-128): p += eps/2 g; q += eps p; g = grad log pi(q);
// p += eps/2 g, momenta ~ N(0, I), Metropolis accept on the energy difference.
// Momenta and the accept uniform are counter-based Philox4x32-10 draws keyed by
// (seed; step, chain, element): every rank of a row-sharded run draws the same
// numbers, so replicated chains need no broadcast.
//
// One thread per chain; parameter vectors are tiny ([C, P] with P <= a few hundred)
// and stay resident in device memory between the launches of a trajectory.
#include "bvi_common.h"

namespace {

struct Blocks {
  int n;
  bvi_param_block b[BVI_MAX_PARAM_BLOCKS];
};

__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ int usize_of(const bvi_param_block& b) {
  return b.bijector == BVI_BIJ_SOFTMAX_CENTERED ? b.size - 1 : b.size;
}

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

__global__ void philox_raw_kernel(uint64_t seed, uint64_t step, uint32_t stream, int n_chains, int n_blocks,
                                  uint32_t* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_chains * n_blocks) return;
  const U4 r = draw(seed, step, idx / n_blocks, idx % n_blocks, stream);
  out[4 * idx] = r.x; out[4 * idx + 1] = r.y; out[4 * idx + 2] = r.z; out[4 * idx + 3] = r.w;
}

__global__ void philox_normal_kernel(uint64_t seed, uint64_t step, int C, int P, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  for (int b = 0; 4 * b < P; ++b) {
    float v[4];
    normal4(draw(seed, step, c, b, 0), v);
    for (int e = 0; e < 4 && 4 * b + e < P; ++e) out[(int64_t)c * P + 4 * b + e] = v[e];
  }
}

// ---- posterior assembly ----
// theta = f(z)
__global__ void constrain_kernel(Blocks bl, int C, int P, int Pu, const float* __restrict__ z,
                                 float* __restrict__ theta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  constrain(bl, z + (int64_t)c * Pu, theta + (int64_t)c * P);
}

// lp = ll + log prior(theta) + fldj(z); gz = chain rule of (gl + grad prior); gl is overwritten
__global__ void assemble_kernel(Blocks bl, int C, int P, int Pu, const float* __restrict__ z,
                                float* __restrict__ theta, const float* __restrict__ ll, float* __restrict__ gl,
                                float* __restrict__ lp, float* __restrict__ gz) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float* zc = z + (int64_t)c * Pu;
  float* th = theta + (int64_t)c * P;
  const float fldj = constrain(bl, zc, th);
  float* g = gl ? gl + (int64_t)c * P : nullptr;
  const float lpr = log_prior(bl, th, g);
  lp[c] = ll[c] + lpr + fldj;
  if (g && gz) chain_rule(bl, zc, th, g, gz + (int64_t)c * Pu);
}

// leapfrog "pre": p += eps/2 g; z += eps p; theta = f(z)
__global__ void leapfrog_pre_kernel(Blocks bl, int C, int P, int Pu, float eps, float* __restrict__ z,
                                    float* __restrict__ p, const float* __restrict__ g,
                                    float* __restrict__ theta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float* zc = z + (int64_t)c * Pu;
  float* pc = p + (int64_t)c * Pu;
  const float* gc = g + (int64_t)c * Pu;
  for (int j = 0; j < Pu; ++j) {
    const float pj = pc[j] + 0.5f * eps * gc[j];
    pc[j] = pj;
    zc[j] += eps * pj;
  }
  constrain(bl, zc, theta + (int64_t)c * P);
}

// leapfrog "post": assemble lp, g at the new z, then p += eps/2 g
__global__ void leapfrog_post_kernel(Blocks bl, int C, int P, int Pu, float eps, const float* __restrict__ z,
                                     float* __restrict__ p, float* __restrict__ theta,
                                     const float* __restrict__ ll, float* __restrict__ gl,
                                     float* __restrict__ lp, float* __restrict__ g) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float* zc = z + (int64_t)c * Pu;
  float* th = theta + (int64_t)c * P;
  const float fldj = constrain(bl, zc, th);
  float* glc = gl + (int64_t)c * P;
  const float lpr = log_prior(bl, th, glc);
  lp[c] = ll[c] + lpr + fldj;
  float* gc = g + (int64_t)c * Pu;
  chain_rule(bl, zc, th, glc, gc);
  float* pc = p + (int64_t)c * Pu;
  for (int j = 0; j < Pu; ++j) pc[j] += 0.5f * eps * gc[j];
}

// HMC begin: save the current state, draw (or copy) momenta, record the initial kinetic energy
__global__ void hmc_begin_kernel(int C, int Pu, uint64_t seed, uint64_t step, const float* __restrict__ z,
                                 const float* __restrict__ g, const float* __restrict__ lp,
                                 const float* __restrict__ momenta_in, float* __restrict__ p,
                                 float* __restrict__ z0, float* __restrict__ g0, float* __restrict__ lp0,
                                 float* __restrict__ k0) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float ke = 0.f;
  for (int b = 0; 4 * b < Pu; ++b) {
    float v[4];
    if (!momenta_in) normal4(draw(seed, step, c, b, 0), v);
    for (int e = 0; e < 4 && 4 * b + e < Pu; ++e) {
      const int64_t i = (int64_t)c * Pu + 4 * b + e;
      const float pj = momenta_in ? momenta_in[i] : v[e];
      p[i] = pj;
      ke += 0.5f * pj * pj;
      z0[i] = z[i];
      g0[i] = g[i];
    }
  }
  lp0[c] = lp[c];
  k0[c] = ke;
}

// HMC end: Metropolis accept on the energy difference; rejected or non-finite proposals restore the state
__global__ void hmc_end_kernel(int C, int Pu, uint64_t seed, uint64_t step, const float* __restrict__ log_u_in,
                               float* __restrict__ z, float* __restrict__ g, float* __restrict__ lp,
                               const float* __restrict__ p, const float* __restrict__ z0,
                               const float* __restrict__ g0, const float* __restrict__ lp0,
                               const float* __restrict__ k0, unsigned char* __restrict__ accepted,
                               float* __restrict__ log_ratio_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float k1 = 0.f;
  for (int j = 0; j < Pu; ++j) {
    const float pj = p[(int64_t)c * Pu + j];
    k1 += 0.5f * pj * pj;
  }
  float lr = (lp[c] - k1) - (lp0[c] - k0[c]);
  if (!isfinite(lr)) lr = -INFINITY;
  const float lu = log_u_in ? log_u_in[c] : logf(u01(draw(seed, step, c, 0, 1).x));
  const bool acc = lu < lr;
  if (!acc) {
    for (int j = 0; j < Pu; ++j) {
      z[(int64_t)c * Pu + j] = z0[(int64_t)c * Pu + j];
      g[(int64_t)c * Pu + j] = g0[(int64_t)c * Pu + j];
    }
    lp[c] = lp0[c];
  }
  accepted[c] = acc ? 1 : 0;
  log_ratio_out[c] = lr;
}

int make_blocks(const bvi_param_block* blocks, int n_blocks, Blocks* out, int* P, int* Pu) {
  if (!blocks || n_blocks < 1 || n_blocks > BVI_MAX_PARAM_BLOCKS) return BVI_ERR_ARG;
  out->n = n_blocks;
  *P = 0;
  *Pu = 0;
  for (int i = 0; i < n_blocks; ++i) {
    const bvi_param_block b = blocks[i];
    if (b.size < 1 || b.bijector < BVI_BIJ_IDENTITY || b.bijector > BVI_BIJ_SOFTMAX_CENTERED ||
        b.prior < BVI_PRIOR_NONE || b.prior > BVI_PRIOR_DIRICHLET)
      return BVI_ERR_ARG;
    if (b.bijector == BVI_BIJ_SOFTMAX_CENTERED && b.size < 2) return BVI_ERR_ARG;
    out->b[i] = b;
    *P += b.size;
    *Pu += b.bijector == BVI_BIJ_SOFTMAX_CENTERED ? b.size - 1 : b.size;
  }
  return BVI_OK;
}

inline dim3 chain_grid(int C) { return dim3(bvi_ceil_div(C, 64)); }

}  // namespace

extern "C" {

int bvi_philox_uint32(uint64_t seed, uint64_t step, int stream_id, int n_chains, int n_blocks, uint32_t* out,
                      void* stream) {
  if (!out || n_chains < 1 || n_blocks < 1 || stream_id < 0 || stream_id > 15) return BVI_ERR_ARG;
  const int tot = n_chains * n_blocks;
  hipLaunchKernelGGL(philox_raw_kernel, dim3(bvi_ceil_div(tot, 256)), dim3(256), 0, (hipStream_t)stream, seed, step,
                     (uint32_t)stream_id, n_chains, n_blocks, out);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_philox_normal(uint64_t seed, uint64_t step, int n_chains, int n_params, float* out, void* stream) {
  if (!out || n_chains < 1 || n_params < 1) return BVI_ERR_ARG;
  hipLaunchKernelGGL(philox_normal_kernel, chain_grid(n_chains), dim3(64), 0, (hipStream_t)stream, seed, step,
                     n_chains, n_params, out);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_constrain(const bvi_param_block* blocks, int n_blocks, int n_chains, const float* z, float* theta,
                  void* stream) {
  Blocks bl;
  int P, Pu;
  const int st = make_blocks(blocks, n_blocks, &bl, &P, &Pu);
  if (st != BVI_OK) return st;
  if (!z || !theta || n_chains < 1) return BVI_ERR_ARG;
  hipLaunchKernelGGL(constrain_kernel, chain_grid(n_chains), dim3(64), 0, (hipStream_t)stream, bl, n_chains, P, Pu,
                     z, theta);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_posterior_assemble(const bvi_param_block* blocks, int n_blocks, int n_chains, const float* z, float* theta,
                           const float* loglik, float* grad_theta, float* lp, float* grad_z, void* stream) {
  Blocks bl;
  int P, Pu;
  const int st = make_blocks(blocks, n_blocks, &bl, &P, &Pu);
  if (st != BVI_OK) return st;
  if (!z || !theta || !loglik || !lp || n_chains < 1 || (grad_z && !grad_theta)) return BVI_ERR_ARG;
  hipLaunchKernelGGL(assemble_kernel, chain_grid(n_chains), dim3(64), 0, (hipStream_t)stream, bl, n_chains, P, Pu,
                     z, theta, loglik, grad_theta, lp, grad_z);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_leapfrog_pre(const bvi_param_block* blocks, int n_blocks, int n_chains, float step_size, float* z, float* p,
                     const float* grad_z, float* theta, void* stream) {
  Blocks bl;
  int P, Pu;
  const int st = make_blocks(blocks, n_blocks, &bl, &P, &Pu);
  if (st != BVI_OK) return st;
  if (!z || !p || !grad_z || !theta || n_chains < 1) return BVI_ERR_ARG;
  hipLaunchKernelGGL(leapfrog_pre_kernel, chain_grid(n_chains), dim3(64), 0, (hipStream_t)stream, bl, n_chains, P,
                     Pu, step_size, z, p, grad_z, theta);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_leapfrog_post(const bvi_param_block* blocks, int n_blocks, int n_chains, float step_size, const float* z,
                      float* p, float* theta, const float* loglik, float* grad_theta, float* lp, float* grad_z,
                      void* stream) {
  Blocks bl;
  int P, Pu;
  const int st = make_blocks(blocks, n_blocks, &bl, &P, &Pu);
  if (st != BVI_OK) return st;
  if (!z || !p || !theta || !loglik || !grad_theta || !lp || !grad_z || n_chains < 1) return BVI_ERR_ARG;
  hipLaunchKernelGGL(leapfrog_post_kernel, chain_grid(n_chains), dim3(64), 0, (hipStream_t)stream, bl, n_chains, P,
                     Pu, step_size, z, p, theta, loglik, grad_theta, lp, grad_z);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_hmc_begin(int n_chains, int n_unconstrained, uint64_t seed, uint64_t step, const float* z,
                  const float* grad_z, const float* lp, const float* momenta_in, float* p, float* z0, float* g0,
                  float* lp0, float* k0, void* stream) {
  if (!z || !grad_z || !lp || !p || !z0 || !g0 || !lp0 || !k0 || n_chains < 1 || n_unconstrained < 1)
    return BVI_ERR_ARG;
  hipLaunchKernelGGL(hmc_begin_kernel, chain_grid(n_chains), dim3(64), 0, (hipStream_t)stream, n_chains,
                     n_unconstrained, seed, step, z, grad_z, lp, momenta_in, p, z0, g0, lp0, k0);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_hmc_end(int n_chains, int n_unconstrained, uint64_t seed, uint64_t step, const float* log_u_in, float* z,
                float* grad_z, float* lp, const float* p, const float* z0, const float* g0, const float* lp0,
                const float* k0, unsigned char* is_accepted, float* log_accept_ratio, void* stream) {
  if (!z || !grad_z || !lp || !p || !z0 || !g0 || !lp0 || !k0 || !is_accepted || !log_accept_ratio ||
      n_chains < 1 || n_unconstrained < 1)
    return BVI_ERR_ARG;
  hipLaunchKernelGGL(hmc_end_kernel, chain_grid(n_chains), dim3(64), 0, (hipStream_t)stream, n_chains,
                     n_unconstrained, seed, step, log_u_in, z, grad_z, lp, p, z0, g0, lp0, k0, is_accepted,
                     log_accept_ratio);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

}  // extern "C"
