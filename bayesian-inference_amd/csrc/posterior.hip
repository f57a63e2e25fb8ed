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
// One workgroup per chain (64 threads for the usual handful of parameters, up to 1024 for the tens of
// thousands of a full-covariance mixture); the [C, P] vectors stay resident in device memory between
// the launches of a trajectory.
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

// ---- posterior assembly: one workgroup per chain, threads stride over the parameter elements ----
// log prior of one element and its score
__device__ __forceinline__ float prior_elem(const bvi_param_block& b, int j, float th, float* score) {
  if (b.prior == BVI_PRIOR_NORMAL) {
    const float u = (th - b.a) / b.b;
    *score = -u / b.b;
    return -0.5f * u * u - 0.5f * BVI_LN2PI - logf(b.b);
  }
  if (b.prior == BVI_PRIOR_HALFNORMAL) {
    const float u = th / b.a;
    *score = -u / b.a;
    return th < 0.f ? -INFINITY : (0.5f * logf(2.f / 3.14159265358979f) - logf(b.a) - 0.5f * u * u);
  }
  if (b.prior == BVI_PRIOR_DIRICHLET) {
    *score = (b.a - 1.f) / th;
    return (b.a - 1.f) * logf(th) + (j == 0 ? lgammaf(b.a * b.size) - b.size * lgammaf(b.a) : 0.f);
  }
  if (b.prior == BVI_PRIOR_WISHART_TRIL) {
    // log Wishart(L L^T; df = a, V = b^2 I) for a Cholesky factor L, no Jacobian of L -> L L^T (TFP's
    // input_output_cholesky=True):  (a - d - 1) sum_i log L_ii - |L|_F^2 / (2 b^2) - log Z, element by element;
    // log Z = a d log b + a d log(2) / 2 + d (d - 1) log(pi) / 4 + sum_{m < d} lgamma((a - m) / 2), once per matrix
    const int d = b.dim, f = j % (d * d), row = f / d, col = f - row * d;
    float lp = 0.f;
    *score = 0.f;
    if (col <= row) {
      lp = -0.5f * th * th / (b.b * b.b);
      *score = -th / (b.b * b.b);
      if (col == row) {
        lp += (b.a - (float)d - 1.f) * logf(th);
        *score += (b.a - (float)d - 1.f) / th;
      }
    }
    if (f == 0) {
      float z = b.a * (float)d * (logf(b.b) + 0.5f * 0.69314718056f) + 0.25f * (float)(d * (d - 1)) * 1.1447298858f;
      for (int m = 0; m < d; ++m) z += lgammaf(0.5f * (b.a - (float)m));
      lp -= z;
    }
    return lp;
  }
  *score = 0.f;
  return 0.f;
}

// The work of one chain (one workgroup; `red` = __shared__ float[16] for its reductions).
// PRE: the position the constraint is applied to is
// z + eps (p + eps/2 g), formed on the fly from the old z, p, g (which are updated in place afterwards).
// ASSEMBLE: also priors, fldj, chain rule (gl gets the prior score added in place, gz is written);
// HALFSTEP: then p += eps/2 gz.  Every in-place update is made by the lane that owns the element,
// after all lanes have read what they need of it.
template <bool PRE, bool ASSEMBLE, bool HALFSTEP>
__device__ void chain_work(const Blocks& bl, int P, int Pu, float eps, float* __restrict__ z, float* __restrict__ p,
                           const float* __restrict__ g_old, float* __restrict__ theta, const float* __restrict__ ll,
                           float* __restrict__ gl, float* __restrict__ lp, float* __restrict__ gz, int c,
                           float* red) {
  const int lane = threadIdx.x, nt = blockDim.x;
  auto zval = [&](int i) -> float {
    if (PRE) return fmaf(eps, fmaf(0.5f * eps, g_old[i], p[i]), z[i]);
    return z[i];
  };
  float part = 0.f;      // this lane's share of log prior + fldj
  // All blocks whose bijector acts element by element (identity, softplus, exp) in ONE sweep over the flat
  // parameter vector -- a thread finds the block of its element by a scan of the (at most 16) block sizes --
  // so that the loads of the whole vector are in flight together instead of block after block.
  for (int t = lane; t < P; t += nt) {
    int i = 0, bo_t = 0, bo_z = 0;
    for (; i < bl.n - 1; ++i) {
      if (t < bo_t + bl.b[i].size) break;
      bo_t += bl.b[i].size;
      bo_z += usize_of(bl.b[i]);
    }
    const bvi_param_block b = bl.b[i];
    if (b.bijector == BVI_BIJ_SOFTMAX_CENTERED || b.bijector == BVI_BIJ_FILL_SCALE_TRIL) continue;
    const int zi = bo_z + (t - bo_t);
    const float x = zval(zi);
    float th, fl = 0.f, dth = 1.f, dfl = 0.f;   // theta, fldj share, d theta / dx, d fldj / dx
    if (b.bijector == BVI_BIJ_SOFTPLUS) {
      th = softplusf(x); fl = -softplusf(-x); dth = sigmoidf(x); dfl = sigmoidf(-x);
    } else if (b.bijector == BVI_BIJ_EXP) {
      th = expf(x); fl = x; dth = th; dfl = 1.f;
    } else {
      th = x;
    }
    theta[t] = th;
    if (ASSEMBLE) {
      float sc;
      part += fl + prior_elem(b, t - bo_t, th, &sc);
      if (gl) {
        const float gt = gl[t] + sc;
        gl[t] = gt;
        if (gz) {
          const float gzj = fmaf(gt, dth, dfl);
          gz[zi] = gzj;
          if (HALFSTEP) p[zi] = fmaf(0.5f * eps, gzj, p[zi]);
        }
      }
    }
    if (PRE) {         // this thread owns the element: update it in place
      const float pj = fmaf(0.5f * eps, g_old[zi], p[zi]);
      p[zi] = pj;
      z[zi] = x;
    }
  }
  // blocks that need the whole block at once (softmax: reductions; triangular fill: index map)
  int oz = 0, ot = 0;
  for (int i = 0; i < bl.n; ++i) {
    const bvi_param_block b = bl.b[i];
    if (b.bijector == BVI_BIJ_SOFTMAX_CENTERED) {
      float m = 0.f;     // the appended logit is 0
      for (int j = lane; j < b.size - 1; j += nt) m = fmaxf(m, zval(oz + j));
      m = block_reduce<true>(m, red);
      float sp = 0.f;
      for (int j = lane; j < b.size - 1; j += nt) sp += expf(zval(oz + j) - m);
      const float lse = m + logf(expf(-m) + block_reduce<false>(sp, red));
      float inner_p = 0.f;
      for (int j = lane; j < b.size; j += nt) {
        const float ly = (j < b.size - 1 ? zval(oz + j) : 0.f) - lse;
        const float th = expf(ly);
        theta[ot + j] = th;
        if (ASSEMBLE) {
          float sc;
          part += ly + (j == 0 ? 0.5f * logf((float)b.size) : 0.f) + prior_elem(b, j, th, &sc);
          if (gl) {
            const float gt = gl[ot + j] + sc;
            gl[ot + j] = gt;
            inner_p += th * gt;
          }
        }
      }
      if (ASSEMBLE && gl && gz) {
        const float inner = block_reduce<false>(inner_p, red);
        for (int j = lane; j < b.size - 1; j += nt) {
          const float th = expf(zval(oz + j) - lse);
          const float gzj = th * (gl[ot + j] - inner) + 1.f - (float)b.size * th;
          gz[oz + j] = gzj;
          if (HALFSTEP) p[oz + j] = fmaf(0.5f * eps, gzj, p[oz + j]);
        }
      }
      if (PRE) {
        __syncthreads();   // every lane has read the old z, p of this block; now update them in place
        for (int j = lane; j < b.size - 1; j += nt) {
          const float pj = fmaf(0.5f * eps, g_old[oz + j], p[oz + j]);
          p[oz + j] = pj;
          z[oz + j] = fmaf(eps, pj, z[oz + j]);
        }
      }
    } else if (b.bijector == BVI_BIJ_FILL_SCALE_TRIL) {
      // lanes stride over the constrained entries; every lower-triangular entry reads (and, in place,
      // updates) its own packed element, entries above the diagonal are structural zeros
      const int d = b.dim, dd = d * d, m = d * (d + 1) / 2;
      for (int t = lane; t < b.size; t += nt) {
        const int mat = t / dd, f = t - mat * dd, i = f / d, j = f - i * d;
        float th = 0.f, fl = 0.f, dth = 1.f, dfl = 0.f, x = 0.f;
        const int src = oz + mat * m + fill_triangular_src(d, f);
        if (j <= i) {
          x = zval(src);
          th = x;
          if (i == j) {
            th = softplusf(x) + b.shift; fl = -softplusf(-x); dth = sigmoidf(x); dfl = sigmoidf(-x);
          }
        }
        theta[ot + t] = th;
        if (ASSEMBLE) {
          float sc;
          part += fl + prior_elem(b, t, th, &sc);
          if (gl) {
            const float gt = gl[ot + t] + sc;
            gl[ot + t] = gt;
            if (gz && j <= i) {
              const float gzj = fmaf(gt, dth, dfl);
              gz[src] = gzj;
              if (HALFSTEP) p[src] = fmaf(0.5f * eps, gzj, p[src]);
            }
          }
        }
        if (PRE && j <= i) {
          const float pj = fmaf(0.5f * eps, g_old[src], p[src]);
          p[src] = pj;
          z[src] = x;
        }
      }
    }
    oz += usize_of(b);
    ot += b.size;
  }
  if (ASSEMBLE) {
    part = block_reduce<false>(part, red);
    if (lane == 0) lp[c] = ll[c] + part;
  }
}

// theta = f(z)
__global__ __launch_bounds__(1024) void constrain_kernel(Blocks bl, int C, int P, int Pu, const float* __restrict__ z,
                                                       float* __restrict__ theta) {
  const int c = blockIdx.x;
  __shared__ float red[16];
  chain_work<false, false, false>(bl, P, Pu, 0.f, const_cast<float*>(z) + (int64_t)c * Pu, nullptr, nullptr,
                                  theta + (int64_t)c * P, nullptr, nullptr, nullptr, nullptr, c, red);
}

// lp = ll + log prior(theta) + fldj(z); gz = chain rule of (gl + grad prior); gl is overwritten
__global__ __launch_bounds__(1024) void assemble_kernel(Blocks bl, int C, int P, int Pu, const float* __restrict__ z,
                                                      float* __restrict__ theta, const float* __restrict__ ll,
                                                      float* __restrict__ gl, float* __restrict__ lp,
                                                      float* __restrict__ gz) {
  const int c = blockIdx.x;
  __shared__ float red[16];
  chain_work<false, true, false>(bl, P, Pu, 0.f, const_cast<float*>(z) + (int64_t)c * Pu, nullptr, nullptr,
                                 theta + (int64_t)c * P, ll, gl ? gl + (int64_t)c * P : nullptr, lp,
                                 gz ? gz + (int64_t)c * Pu : nullptr, c, red);
}

// leapfrog "pre": p += eps/2 g; z += eps p; theta = f(z)
__global__ __launch_bounds__(1024) void leapfrog_pre_kernel(Blocks bl, int C, int P, int Pu, float eps,
                                                          const float* __restrict__ eps_chain,
                                                          float* __restrict__ z, float* __restrict__ p,
                                                          const float* __restrict__ g, float* __restrict__ theta) {
  const int c = blockIdx.x;
  __shared__ float red[16];
  if (eps_chain) {
    eps = eps_chain[c];                   // per-chain signed step; 0 = frozen chain: only theta = f(z) is rewritten
    if (eps == 0.f) {
      chain_work<false, false, false>(bl, P, Pu, 0.f, z + (int64_t)c * Pu, nullptr, nullptr, theta + (int64_t)c * P,
                                      nullptr, nullptr, nullptr, nullptr, c, red);
      return;
    }
  }
  chain_work<true, false, false>(bl, P, Pu, eps, z + (int64_t)c * Pu, p + (int64_t)c * Pu, g + (int64_t)c * Pu,
                                 theta + (int64_t)c * P, nullptr, nullptr, nullptr, nullptr, c, red);
}

// leapfrog "post": assemble lp, g at the new z, then p += eps/2 g
__global__ __launch_bounds__(1024) void leapfrog_post_kernel(Blocks bl, int C, int P, int Pu, float eps,
                                                           const float* __restrict__ eps_chain,
                                                           const float* __restrict__ z, float* __restrict__ p,
                                                           float* __restrict__ theta, const float* __restrict__ ll,
                                                           float* __restrict__ gl, float* __restrict__ lp,
                                                           float* __restrict__ g) {
  const int c = blockIdx.x;
  __shared__ float red[16];
  if (eps_chain) {
    eps = eps_chain[c];
    if (eps == 0.f) return;               // frozen chain: momentum, log density and gradient stay as they are
  }
  chain_work<false, true, true>(bl, P, Pu, eps, const_cast<float*>(z) + (int64_t)c * Pu, p + (int64_t)c * Pu,
                                nullptr, theta + (int64_t)c * P, ll, gl + (int64_t)c * P, lp, g + (int64_t)c * Pu, c, red);
}

// leapfrog "post" of one step and "pre" of the next in one launch (interior steps of a trajectory):
// lp, g = assemble at z; p += eps g (two half steps); z += eps p; theta = f(z)
__global__ __launch_bounds__(1024) void leapfrog_mid_kernel(Blocks bl, int C, int P, int Pu, float eps,
                                                            float* __restrict__ z, float* __restrict__ p,
                                                            float* __restrict__ theta, const float* __restrict__ ll,
                                                            float* __restrict__ gl, float* __restrict__ lp,
                                                            float* __restrict__ g) {
  const int c = blockIdx.x;
  __shared__ float red[16];
  chain_work<false, true, true>(bl, P, Pu, eps, z + (int64_t)c * Pu, p + (int64_t)c * Pu, nullptr,
                                theta + (int64_t)c * P, ll, gl + (int64_t)c * P, lp, g + (int64_t)c * Pu, c, red);
  __syncthreads();     // the gradient and momentum written above are read back below (by the same threads)
  chain_work<true, false, false>(bl, P, Pu, eps, z + (int64_t)c * Pu, p + (int64_t)c * Pu, g + (int64_t)c * Pu,
                                 theta + (int64_t)c * P, nullptr, nullptr, nullptr, nullptr, c, red);
}

// HMC begin: save the current state, draw (or copy) momenta, record the initial kinetic energy.
// One workgroup per chain; a thread owns whole Philox blocks (4 consecutive elements).
__global__ __launch_bounds__(1024) void hmc_begin_kernel(int C, int Pu, uint64_t seed, uint64_t step,
                                                         const float* __restrict__ z, const float* __restrict__ g,
                                                         const float* __restrict__ lp,
                                                         const float* __restrict__ momenta_in, float* __restrict__ p,
                                                         float* __restrict__ z0, float* __restrict__ g0,
                                                         float* __restrict__ lp0, float* __restrict__ k0,
                                                         const uint64_t* __restrict__ step_offset) {
  const int c = blockIdx.x;
  __shared__ float red[16];
  if (step_offset) step += *step_offset;
  float ke = 0.f;
  for (int b = threadIdx.x; 4 * b < Pu; b += blockDim.x) {
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
  ke = block_reduce<false>(ke, red);
  if (threadIdx.x == 0) {
    lp0[c] = lp[c];
    k0[c] = ke;
  }
}

// HMC end: Metropolis accept on the energy difference; rejected or non-finite proposals restore the state
__global__ __launch_bounds__(1024) void hmc_end_kernel(int C, int Pu, uint64_t seed, uint64_t step,
                                                       const float* __restrict__ log_u_in, float* __restrict__ z,
                                                       float* __restrict__ g, float* __restrict__ lp,
                                                       const float* __restrict__ p, const float* __restrict__ z0,
                                                       const float* __restrict__ g0, const float* __restrict__ lp0,
                                                       const float* __restrict__ k0,
                                                       unsigned char* __restrict__ accepted,
                                                       float* __restrict__ log_ratio_out,
                                                       const uint64_t* __restrict__ step_offset) {
  const int c = blockIdx.x;
  __shared__ float red[16];
  if (step_offset) step += *step_offset;
  float k1 = 0.f;
  for (int j = threadIdx.x; j < Pu; j += blockDim.x) {
    const float pj = p[(int64_t)c * Pu + j];
    k1 += 0.5f * pj * pj;
  }
  k1 = block_reduce<false>(k1, red);
  // the energy difference in double: log-densities are stored relative to a level that keeps them small
  // (bvi_logp_rebase), so the two differences below are exact and so is their sum to float rounding
  float lr = (float)(((double)lp[c] - (double)lp0[c]) - ((double)k1 - (double)k0[c]));
  if (!isfinite(lr)) lr = -INFINITY;
  const float lu = log_u_in ? log_u_in[c] : logf(u01(draw(seed, step, c, 0, 1).x));
  const bool acc = lu < lr;
  __syncthreads();                       // every thread has read lp[c] before it may be overwritten
  if (!acc) {
    for (int j = threadIdx.x; j < Pu; j += blockDim.x) {
      z[(int64_t)c * Pu + j] = z0[(int64_t)c * Pu + j];
      g[(int64_t)c * Pu + j] = g0[(int64_t)c * Pu + j];
    }
    if (threadIdx.x == 0) lp[c] = lp0[c];
  }
  if (threadIdx.x == 0) {
    accepted[c] = acc ? 1 : 0;
    log_ratio_out[c] = lr;
  }
}

// Re-levels the stored log-densities of all chains: m = the largest finite lp[c]; lp[c] -= m; *ref += m * share.
// With ref handed to the likelihood entry points as logp_ref, every later evaluation comes out relative to the new
// level, i.e. small wherever the chains are, and differences between a proposal and the current state keep float
// precision however large the log-density itself is (at 10M rows |log pi| ~ 1e7, where floats are 1 apart).
// `share` is this row shard's part of the total level (1 / shards for equal shards, rows_local / rows_total in
// general: a shard's log-likelihood is that part of the total up to O(sqrt(rows)) fluctuations, so its own
// level stays next to its own values).  The total level is *total when given (it moves by m on every rank), else
// *ref / share.  lp_abs (optional) receives the absolute values, float(lp[c] + total level), as they were before.
__global__ __launch_bounds__(1024) void logp_rebase_kernel(int C, float* __restrict__ lp, double share,
                                                           double* __restrict__ ref, double* __restrict__ total,
                                                           float* __restrict__ lp_abs) {
  __shared__ float red[16];
  const double level = total ? *total : *ref / share;
  float m = -INFINITY;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float v = lp[c];
    if (isfinite(v)) m = fmaxf(m, v);
  }
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();                       // also: every thread has read the levels before thread 0 rewrites them
  m = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, red[w]);
  if (!isfinite(m)) m = 0.f;             // no chain with a finite density: nothing to level
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float v = lp[c];
    if (lp_abs) lp_abs[c] = (float)((double)v + level);
    lp[c] = v - m;
  }
  if (threadIdx.x == 0) {
    *ref += (double)m * share;
    if (total) *total += (double)m;
  }
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
  const int st = make_blocks(blocks, n_blocks, &bl, &P, &Pu, true);
  if (st != BVI_OK) return st;
  if (!z || !theta || n_chains < 1) return BVI_ERR_ARG;
  hipLaunchKernelGGL(constrain_kernel, dim3(n_chains), dim3(chain_threads(P)), 0, (hipStream_t)stream, bl, n_chains, P, Pu,
                     z, theta);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_posterior_assemble(const bvi_param_block* blocks, int n_blocks, int n_chains, const float* z, float* theta,
                           const float* loglik, float* grad_theta, float* lp, float* grad_z, void* stream) {
  Blocks bl;
  int P, Pu;
  const int st = make_blocks(blocks, n_blocks, &bl, &P, &Pu, true);
  if (st != BVI_OK) return st;
  if (!z || !theta || !loglik || !lp || n_chains < 1 || (grad_z && !grad_theta)) return BVI_ERR_ARG;
  hipLaunchKernelGGL(assemble_kernel, dim3(n_chains), dim3(chain_threads(P)), 0, (hipStream_t)stream, bl, n_chains, P, Pu,
                     z, theta, loglik, grad_theta, lp, grad_z);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

static int launch_pre(const bvi_param_block* blocks, int n_blocks, int n_chains, float step_size,
                      const float* step_sizes, float* z, float* p, const float* grad_z, float* theta, void* stream) {
  Blocks bl;
  int P, Pu;
  const int st = make_blocks(blocks, n_blocks, &bl, &P, &Pu, true);
  if (st != BVI_OK) return st;
  if (!z || !p || !grad_z || !theta || n_chains < 1) return BVI_ERR_ARG;
  hipLaunchKernelGGL(leapfrog_pre_kernel, dim3(n_chains), dim3(chain_threads(P)), 0, (hipStream_t)stream, bl, n_chains, P,
                     Pu, step_size, step_sizes, z, p, grad_z, theta);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

static int launch_post(const bvi_param_block* blocks, int n_blocks, int n_chains, float step_size,
                       const float* step_sizes, const float* z, float* p, float* theta, const float* loglik,
                       float* grad_theta, float* lp, float* grad_z, void* stream) {
  Blocks bl;
  int P, Pu;
  const int st = make_blocks(blocks, n_blocks, &bl, &P, &Pu, true);
  if (st != BVI_OK) return st;
  if (!z || !p || !theta || !loglik || !grad_theta || !lp || !grad_z || n_chains < 1) return BVI_ERR_ARG;
  hipLaunchKernelGGL(leapfrog_post_kernel, dim3(n_chains), dim3(chain_threads(P)), 0, (hipStream_t)stream, bl, n_chains, P,
                     Pu, step_size, step_sizes, z, p, theta, loglik, grad_theta, lp, grad_z);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_leapfrog_pre(const bvi_param_block* blocks, int n_blocks, int n_chains, float step_size, float* z, float* p,
                     const float* grad_z, float* theta, void* stream) {
  return launch_pre(blocks, n_blocks, n_chains, step_size, nullptr, z, p, grad_z, theta, stream);
}

int bvi_leapfrog_post(const bvi_param_block* blocks, int n_blocks, int n_chains, float step_size, const float* z,
                      float* p, float* theta, const float* loglik, float* grad_theta, float* lp, float* grad_z,
                      void* stream) {
  return launch_post(blocks, n_blocks, n_chains, step_size, nullptr, z, p, theta, loglik, grad_theta, lp, grad_z,
                     stream);
}

int bvi_leapfrog_mid(const bvi_param_block* blocks, int n_blocks, int n_chains, float step_size, float* z, float* p,
                     float* theta, const float* loglik, float* grad_theta, float* lp, float* grad_z, void* stream) {
  Blocks bl;
  int P, Pu;
  const int st = make_blocks(blocks, n_blocks, &bl, &P, &Pu, true);
  if (st != BVI_OK) return st;
  if (!z || !p || !theta || !loglik || !grad_theta || !lp || !grad_z || n_chains < 1) return BVI_ERR_ARG;
  hipLaunchKernelGGL(leapfrog_mid_kernel, dim3(n_chains), dim3(chain_threads(P)), 0, (hipStream_t)stream, bl, n_chains,
                     P, Pu, step_size, z, p, theta, loglik, grad_theta, lp, grad_z);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_leapfrog_pre_chainwise(const bvi_param_block* blocks, int n_blocks, int n_chains, const float* step_sizes,
                               float* z, float* p, const float* grad_z, float* theta, void* stream) {
  if (!step_sizes) return BVI_ERR_ARG;
  return launch_pre(blocks, n_blocks, n_chains, 0.f, step_sizes, z, p, grad_z, theta, stream);
}

int bvi_leapfrog_post_chainwise(const bvi_param_block* blocks, int n_blocks, int n_chains, const float* step_sizes,
                                const float* z, float* p, float* theta, const float* loglik, float* grad_theta,
                                float* lp, float* grad_z, void* stream) {
  if (!step_sizes) return BVI_ERR_ARG;
  return launch_post(blocks, n_blocks, n_chains, 0.f, step_sizes, z, p, theta, loglik, grad_theta, lp, grad_z,
                     stream);
}

int bvi_hmc_begin(int n_chains, int n_unconstrained, uint64_t seed, uint64_t step, const float* z,
                  const float* grad_z, const float* lp, const float* momenta_in, float* p, float* z0, float* g0,
                  float* lp0, float* k0, const uint64_t* step_offset, void* stream) {
  if (!z || !grad_z || !lp || !p || !z0 || !g0 || !lp0 || !k0 || n_chains < 1 || n_unconstrained < 1)
    return BVI_ERR_ARG;
  hipLaunchKernelGGL(hmc_begin_kernel, dim3(n_chains), dim3(chain_threads(n_unconstrained)), 0, (hipStream_t)stream,
                     n_chains,
                     n_unconstrained, seed, step, z, grad_z, lp, momenta_in, p, z0, g0, lp0, k0, step_offset);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_hmc_end(int n_chains, int n_unconstrained, uint64_t seed, uint64_t step, const float* log_u_in, float* z,
                float* grad_z, float* lp, const float* p, const float* z0, const float* g0, const float* lp0,
                const float* k0, unsigned char* is_accepted, float* log_accept_ratio, const uint64_t* step_offset,
                void* stream) {
  if (!z || !grad_z || !lp || !p || !z0 || !g0 || !lp0 || !k0 || !is_accepted || !log_accept_ratio ||
      n_chains < 1 || n_unconstrained < 1)
    return BVI_ERR_ARG;
  hipLaunchKernelGGL(hmc_end_kernel, dim3(n_chains), dim3(chain_threads(n_unconstrained)), 0, (hipStream_t)stream,
                     n_chains,
                     n_unconstrained, seed, step, log_u_in, z, grad_z, lp, p, z0, g0, lp0, k0, is_accepted,
                     log_accept_ratio, step_offset);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_logp_rebase(int n_chains, float* lp, int n_shards, double* logp_ref, float* lp_absolute, void* stream) {
  if (!lp || !logp_ref || n_chains < 1 || n_shards < 1) return BVI_ERR_ARG;
  hipLaunchKernelGGL(logp_rebase_kernel, dim3(1), dim3(n_chains > 256 ? 1024 : 256), 0, (hipStream_t)stream, n_chains,
                     lp, 1.0 / (double)n_shards, logp_ref, (double*)nullptr, lp_absolute);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_logp_rebase_shard(int n_chains, float* lp, double share, double* levels, float* lp_absolute, void* stream) {
  if (!lp || !levels || n_chains < 1 || !(share > 0.0) || !(share <= 1.0)) return BVI_ERR_ARG;
  hipLaunchKernelGGL(logp_rebase_kernel, dim3(1), dim3(n_chains > 256 ? 1024 : 256), 0, (hipStream_t)stream, n_chains,
                     lp, share, levels, levels + 1, lp_absolute);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

}  // extern "C"
