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
#include "posterior_device.h"

namespace {

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
