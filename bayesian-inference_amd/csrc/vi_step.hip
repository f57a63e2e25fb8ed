// The O(samples x parameters) part of one mean-field ADVI optimisation step, around the fused log-posterior pass:
//   z_s = loc + softplus(raw) eps_s                                              (bvi_advi_meanfield_sample)
//   [bvi_constrain -> <family>_logp_grad -> bvi_posterior_assemble give lp_s = log pi(z_s), d lp_s / d z_s]
//   loss = -mean_s (lp_s - log q(z_s)) over the samples whose term is finite,    (bvi_advi_meanfield_grads)
//   d loss / d loc_j = -mean_s d lp_s / d z_sj
//   d loss / d raw_j = -(mean_s (d lp_s / d z_sj) eps_sj + 1 / sigma_j) sigmoid(raw_j)
//   per variable: g <- g clip / max(|g|, clip); Adam update                      (bvi_clip_adam_step)
// Replaces, for this surrogate, what the reference does with a GradientTape over
// tfp.vi.monte_carlo_variational_loss, tf.clip_by_norm per variable and tf.optimizers.Adam
// (bayes_vi/inference/vi/vi.py:55-75, 109-116; surrogate_posteriors.py:113-123): the same numbers as autograd over
// the same formulas, in four small launches instead of some sixty.  log q is the diagonal Gaussian's density in
// unconstrained space; lp_s already contains the bijectors' log-determinant, so lp_s - log q(z_s) equals the
// constrained-space difference the reference forms.
#include "bvi_common.h"

namespace {

__device__ __forceinline__ float vi_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // torch's threshold
__device__ __forceinline__ float vi_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void advi_sample_kernel(int S, int Pu, const float* __restrict__ eps, const float* __restrict__ loc,
                                   const float* __restrict__ raw, float* __restrict__ z) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)S * Pu) return;
  const int j = (int)(i % Pu);
  z[i] = fmaf(vi_softplus(raw[j]), eps[i], loc[j]);
}

__device__ __forceinline__ double block_sum(double v, double* red) {
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();                                  // red may still be read from the previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
  return t;
}

// Which samples count.  The reference drops the samples whose discrepancy is not finite before taking the mean
// (vi.py:55-58); a sample whose log density is finite while an entry of its gradient is not (a scale underflowing
// towards 0: log pi = -1e38, d log pi / d z = -inf) would pass that test and put NaN into the variables for good.
// One predicate for every block of these kernels -- loss and gradients average over the same samples --:
// log density finite and every entry of the sample's gradient finite.  Kept in LDS (one byte per sample) when the
// samples fit, recomputed per use otherwise.
constexpr int VI_MAX_FLAGS = 16384;
__device__ __forceinline__ bool vi_sample_counts(const float* __restrict__ lp, const float* __restrict__ gz, int Pu,
                                                 int s) {
  bool ok = isfinite(lp[s]);
  for (int k = 0; k < Pu; ++k) ok = ok && isfinite(gz[(int64_t)s * Pu + k]);
  return ok;
}
__device__ __forceinline__ void vi_fill_flags(unsigned char* flags, int S, const float* __restrict__ lp,
                                              const float* __restrict__ gz, int Pu) {
  if (S <= VI_MAX_FLAGS)
    for (int s = threadIdx.x; s < S; s += blockDim.x) flags[s] = vi_sample_counts(lp, gz, Pu, s) ? 1 : 0;
  __syncthreads();
}
#define VI_COUNTS(s) (S <= VI_MAX_FLAGS ? flags[s] != 0 : vi_sample_counts(lp, gz, Pu, s))

// block j < Pu: the two gradient entries of parameter j; block Pu: the loss.  Sums over the samples in double, in a
// fixed order.
__global__ __launch_bounds__(256) void advi_grads_kernel(int S, int Pu, const float* __restrict__ lp,
                                                         const float* __restrict__ gz,
                                                         const float* __restrict__ eps,
                                                         const float* __restrict__ raw, float* __restrict__ loss,
                                                         float* __restrict__ g_loc, float* __restrict__ g_raw) {
  __shared__ double red[4];
  __shared__ unsigned char flags[VI_MAX_FLAGS];
  const int j = blockIdx.x;
  double nf = 0.0, a = 0.0, b = 0.0;
  vi_fill_flags(flags, S, lp, gz, Pu);
  if (j < Pu) {
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
      const bool fin = VI_COUNTS(s);
      const float g = fin ? gz[(int64_t)s * Pu + j] : 0.f;
      nf += fin ? 1.0 : 0.0;
      a += (double)g;
      b += (double)g * (double)eps[(int64_t)s * Pu + j];
    }
    nf = block_sum(nf, red); a = block_sum(a, red); b = block_sum(b, red);
    if (threadIdx.x == 0) {
      const float r = raw[j];
      g_loc[j] = (float)(-a / nf);
      g_raw[j] = (float)(-(b / nf + 1.0 / (double)vi_softplus(r)) * (double)vi_sigmoid(r));
    }
  } else {
    // log q(z_s) = sum_j (-eps_sj^2 / 2 - log(2 pi) / 2 - log sigma_j)
    double lsig = 0.0;
    for (int k = 0; k < Pu; ++k) lsig += log((double)vi_softplus(raw[k]));
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
      double e2 = 0.0;
      for (int k = 0; k < Pu; ++k) { const double e = (double)eps[(int64_t)s * Pu + k]; e2 += e * e; }
      const double logq = -0.5 * e2 - 0.5 * Pu * BVI_LN2PI_D - lsig;
      const double d = -((double)lp[s] - logq);
      const bool fin = VI_COUNTS(s) && isfinite(d);
      nf += fin ? 1.0 : 0.0;
      a += fin ? d : 0.0;
    }
    nf = block_sum(nf, red); a = block_sum(a, red);
    if (threadIdx.x == 0) *loss = (float)(a / nf);
  }
}

// ---- full-rank ADVI: z_s = loc + L eps_s, L = tril(raw, -1) + diag(softplus(raw_dd) + 1e-5) (surrogate_posteriors.py:
// 125-134); log q(z_s) = sum_j (-eps_sj^2 / 2 - log(2 pi) / 2) - sum_i log L_ii ----
__device__ __forceinline__ float vi_tril_entry(const float* __restrict__ raw, int Pu, int i, int j) {
  return i == j ? vi_softplus(raw[(int64_t)i * Pu + i]) + 1e-5f : raw[(int64_t)i * Pu + j];
}

__global__ void advi_fullrank_sample_kernel(int S, int Pu, const float* __restrict__ eps, const float* __restrict__ loc,
                                            const float* __restrict__ raw, float* __restrict__ z) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)S * Pu) return;
  const int i = (int)(idx % Pu);
  const float* e = eps + (idx / Pu) * Pu;
  float acc = loc[i];
  for (int j = 0; j <= i; ++j) acc = fmaf(vi_tril_entry(raw, Pu, i, j), e[j], acc);
  z[idx] = acc;
}

// block i < Pu: row i of d loss / d raw_tril and entry i of d loss / d loc; block Pu: the loss.
//   d loss / d L_ij = -mean_s g_si eps_sj (j <= i) - [i == j] / L_ii;  the diagonal passes through softplus
__global__ __launch_bounds__(256) void advi_fullrank_grads_kernel(int S, int Pu, const float* __restrict__ lp,
                                                                  const float* __restrict__ gz,
                                                                  const float* __restrict__ eps,
                                                                  const float* __restrict__ raw,
                                                                  float* __restrict__ loss,
                                                                  float* __restrict__ g_loc,
                                                                  float* __restrict__ g_raw) {
  __shared__ double red[4];
  __shared__ double part[4][64];
  __shared__ unsigned char flags[VI_MAX_FLAGS];
  const int i = blockIdx.x;
  vi_fill_flags(flags, S, lp, gz, Pu);
  if (i < Pu) {
    double nf = 0.0, a = 0.0;
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
      const bool fin = VI_COUNTS(s);
      nf += fin ? 1.0 : 0.0;
      a += fin ? (double)gz[(int64_t)s * Pu + i] : 0.0;
    }
    nf = block_sum(nf, red); a = block_sum(a, red);
    if (threadIdx.x == 0) g_loc[i] = (float)(-a / nf);
    const int jl = threadIdx.x & 63, slice = threadIdx.x >> 6;
    for (int j0 = 0; j0 < Pu; j0 += 64) {
      const int j = j0 + jl;
      double acc = 0.0;
      if (j <= i)
        for (int s = slice; s < S; s += 4)
          if (VI_COUNTS(s)) acc += (double)gz[(int64_t)s * Pu + i] * (double)eps[(int64_t)s * Pu + j];
      __syncthreads();
      part[slice][jl] = acc;
      __syncthreads();
      if (slice == 0 && j < Pu) {
        double g = 0.0;
        if (j <= i) {
          g = -(part[0][jl] + part[1][jl] + part[2][jl] + part[3][jl]) / nf;
          if (j == i) {
            const float r = raw[(int64_t)i * Pu + i];
            g = (g - 1.0 / ((double)vi_softplus(r) + 1e-5)) * (double)vi_sigmoid(r);
          }
        }
        g_raw[(int64_t)i * Pu + j] = (float)g;      // above the diagonal: 0, as autograd through tril() leaves it
      }
    }
  } else {
    double lsig = 0.0;
    for (int k = 0; k < Pu; ++k) lsig += log((double)vi_softplus(raw[(int64_t)k * Pu + k]) + 1e-5);
    double nf = 0.0, a = 0.0;
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
      double e2 = 0.0;
      for (int k = 0; k < Pu; ++k) { const double e = (double)eps[(int64_t)s * Pu + k]; e2 += e * e; }
      const double logq = -0.5 * e2 - 0.5 * Pu * BVI_LN2PI_D - lsig;
      const double d = -((double)lp[s] - logq);
      const bool fin = VI_COUNTS(s) && isfinite(d);
      nf += fin ? 1.0 : 0.0;
      a += fin ? d : 0.0;
    }
    nf = block_sum(nf, red); a = block_sum(a, red);
    if (threadIdx.x == 0) *loss = (float)(a / nf);
  }
}

// one workgroup per variable: clip by norm, then the Keras-formulation Adam update
//   m <- b1 m + (1 - b1) g;  v <- b2 v + (1 - b2) g^2;  lr_t = lr sqrt(1 - b2^t) / (1 - b1^t);  x -= lr_t m / (sqrt(v) + eps)
__global__ __launch_bounds__(256) void clip_adam_kernel(int n, float* __restrict__ x, float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        double* __restrict__ t, int advance, float lr, float b1,
                                                        float b2, float eps, float clip) {
  __shared__ double red[4];
  const double tt = *t + (advance ? 1.0 : 0.0);
  double s2 = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s2 += (double)g[i] * (double)g[i];
  s2 = block_sum(s2, red);                           // its barriers also order the read of *t before the write below
  const float norm = (float)sqrt(s2);
  const float factor = clip > 0.f ? fminf(1.f / (norm + 1e-30f), 1.f / clip) * clip : 1.f;   // clip / max(norm, clip)
  const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, tt)) / (1.0 - pow((double)b1, tt)));
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float gi = g[i] * factor;
    g[i] = gi;                                       // the clipped gradient stays readable (tests, diagnostics)
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    x[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
  if (advance && threadIdx.x == 0) *t = tt;
}

}  // namespace

extern "C" {

int bvi_advi_meanfield_sample(int n_samples, int n_unconstrained, const float* eps, const float* loc,
                              const float* raw_scale, float* z, void* stream) {
  if (!eps || !loc || !raw_scale || !z || n_samples < 1 || n_unconstrained < 1) return BVI_ERR_ARG;
  const int64_t tot = (int64_t)n_samples * n_unconstrained;
  hipLaunchKernelGGL(advi_sample_kernel, dim3((unsigned)bvi_ceil_div(tot, 256)), dim3(256), 0, (hipStream_t)stream,
                     n_samples, n_unconstrained, eps, loc, raw_scale, z);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_advi_meanfield_grads(int n_samples, int n_unconstrained, const float* lp, const float* grad_z,
                             const float* eps, const float* raw_scale, float* loss, float* grad_loc,
                             float* grad_raw_scale, void* stream) {
  if (!lp || !grad_z || !eps || !raw_scale || !loss || !grad_loc || !grad_raw_scale || n_samples < 1 ||
      n_unconstrained < 1)
    return BVI_ERR_ARG;
  hipLaunchKernelGGL(advi_grads_kernel, dim3(n_unconstrained + 1), dim3(256), 0, (hipStream_t)stream, n_samples,
                     n_unconstrained, lp, grad_z, eps, raw_scale, loss, grad_loc, grad_raw_scale);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_advi_fullrank_sample(int n_samples, int n_unconstrained, const float* eps, const float* loc,
                             const float* raw_tril, float* z, void* stream) {
  if (!eps || !loc || !raw_tril || !z || n_samples < 1 || n_unconstrained < 1) return BVI_ERR_ARG;
  const int64_t tot = (int64_t)n_samples * n_unconstrained;
  hipLaunchKernelGGL(advi_fullrank_sample_kernel, dim3((unsigned)bvi_ceil_div(tot, 256)), dim3(256), 0,
                     (hipStream_t)stream, n_samples, n_unconstrained, eps, loc, raw_tril, z);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_advi_fullrank_grads(int n_samples, int n_unconstrained, const float* lp, const float* grad_z,
                            const float* eps, const float* raw_tril, float* loss, float* grad_loc,
                            float* grad_raw_tril, void* stream) {
  if (!lp || !grad_z || !eps || !raw_tril || !loss || !grad_loc || !grad_raw_tril || n_samples < 1 ||
      n_unconstrained < 1)
    return BVI_ERR_ARG;
  hipLaunchKernelGGL(advi_fullrank_grads_kernel, dim3(n_unconstrained + 1), dim3(256), 0, (hipStream_t)stream,
                     n_samples, n_unconstrained, lp, grad_z, eps, raw_tril, loss, grad_loc, grad_raw_tril);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_clip_adam_step(int n, float* x, float* grad, float* m, float* v, double* step_count, int advance_step,
                       float learning_rate, float beta_1, float beta_2, float epsilon, float clip_norm,
                       void* stream) {
  if (!x || !grad || !m || !v || !step_count || n < 1) return BVI_ERR_ARG;
  hipLaunchKernelGGL(clip_adam_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, n, x, grad, m, v, step_count,
                     advance_step, learning_rate, beta_1, beta_2, epsilon, clip_norm);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

}  // extern "C"
