// Whole HMC transitions -- momentum draw, L leapfrog steps with the full-data log-posterior
// gradient, Metropolis accept -- for every chain in ONE launch, any number of transitions per
// launch.  For datasets small enough that a workgroup can re-stream them at every leapfrog step
// (the reference's own experiments: 20 ... a few 10^4 points, README.md:114-120, 747-756), where
// the per-step launches of the batched kernels cost more than the arithmetic.
//
// Same transition as tfp.mcmc.HamiltonianMonteCarlo under TransformedTransitionKernel
// (bayes_vi/inference/mcmc/transition_kernels.py:122-128, mcmc.py:153-156) and as the unfused
// sequence bvi_hmc_begin -> L x (bvi_leapfrog_pre, *_logp_grad, bvi_leapfrog_post) -> bvi_hmc_end,
// with the same Philox counters, so both paths draw identical momenta and accept uniforms.
//
// One workgroup per chain.  Lane j of every wave keeps element j of position, momentum and
// gradient in registers for the whole launch (all waves hold identical copies, so no broadcast is
// needed); theta = f(z) goes through LDS to the data loop, in which the workgroup's threads stride
// over the rows with private accumulators; wave butterflies + one LDS exchange give every thread
// the raw sums; lane j turns them into d log lik / d theta_j, adds its prior score, and applies the
// bijector chain rule.  Nothing leaves the chip between the first and the last transition except
// the optional per-transition outputs.
//
// The same kernel runs the No-U-Turn sampler (ALGO = 1): the iterative tree doubling of
// tfp.mcmc.NoUTurnSampler (transition_kernels.py:238-246) -- multinomial sampling inside a subtree,
// biased progressive sampling between subtrees, U-turn checks against checkpoints of sub-tree ends,
// divergence when the energy error exceeds max_energy_diff -- with one chain per workgroup, so that
// the data-dependent tree depth is ordinary uniform control flow.  Same algorithm, step for step, as
// the batched host kernel (inference/mcmc/samplers.py::NoUTurnSamplerKernel); random numbers are
// Philox draws: momenta on stream 0, (direction, subtree-accept) per depth on stream 2, the
// multinomial uniform per leaf on stream 3.  ALGO = 2 is random-walk Metropolis
// (tfp.mcmc.RandomWalkMetropolis, transition_kernels.py:319-321) with the same streams as HMC.
#include "bvi_common.h"
#include "posterior_device.h"

namespace {

constexpr int kMaxP = 64;
constexpr int kMaxQ = 3 * 8 + 1;
constexpr int kMaxDepth = 10;

struct FusedArgs {
  const float* X;
  const float* y;
  int64_t n;
  int d, k;
  int C, P, Pu;
  int T, L;
  float eps;
  const float* eps_chain;
  uint64_t seed, step0;
  int evaluate_initial;
  float* z;
  float* g;
  float* lp;
  const float* momenta_in;
  const float* log_u_in;
  float* z_out;
  float* lp_out;
  unsigned char* acc_out;
  float* lr_out;
  int max_depth;          // NUTS only
  float max_energy_diff;
  int* leap_out;
  unsigned char* div_out;
  float* energy_out;
};

__device__ __forceinline__ float log_add_exp(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + logf(expf(a - m) + expf(b - m));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// FAM: BVI_FAMILY_*; NP: padded feature count (regression) or component count (mixture)
template <int FAM, int NP, int ALGO>
__global__ __launch_bounds__(256) void hmc_fused_kernel(Blocks bl, FusedArgs a) {
  constexpr int Q = FAM == BVI_FAMILY_NORMAL_IID ? 2 : FAM == BVI_FAMILY_LINREG ? NP + 2 : 3 * NP + 1;
  constexpr int PF = FAM == BVI_FAMILY_NORMAL_IID ? 2 : FAM == BVI_FAMILY_LINREG ? NP + 2 : 3 * NP;
  __shared__ float sh_z[kMaxP], sh_theta[kMaxP], sh_gth[kMaxP];
  __shared__ float sh_part[4][kMaxQ];
  __shared__ float sh_ckq[ALGO == 1 ? kMaxDepth + 1 : 1][kMaxP], sh_ckp[ALGO == 1 ? kMaxDepth + 1 : 1][kMaxP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, nwaves = nthreads >> 6;
  const int c = blockIdx.x;
  const int P = a.P, Pu = a.Pu;
  const float nf = (float)a.n;

  // ---- per-lane block metadata: lane j as constrained index (t*) and as unconstrained index (z*) ----
  int t_bij = BVI_BIJ_IDENTITY, t_prior = BVI_PRIOR_NONE, t_size = 1, t_zoff = 0, t_j = 0;
  float t_a = 0.f, t_b = 1.f;
  int z_bij = BVI_BIJ_IDENTITY, z_size = 1, z_toff = 0, z_j = 0;
  {
    int oz = 0, ot = 0;
    for (int i = 0; i < bl.n; ++i) {
      const bvi_param_block b = bl.b[i];
      const int us = usize_of(b);
      if (lane >= ot && lane < ot + b.size) {
        t_bij = b.bijector; t_prior = b.prior; t_size = b.size; t_zoff = oz; t_j = lane - ot; t_a = b.a; t_b = b.b;
      }
      if (lane >= oz && lane < oz + us) {
        z_bij = b.bijector; z_size = b.size; z_toff = ot; z_j = lane - oz;
      }
      oz += us;
      ot += b.size;
    }
  }
  const bool tl = lane < P, zl = lane < Pu;

  float zj = zl ? a.z[(int64_t)c * Pu + lane] : 0.f;
  float gj = zl ? a.g[(int64_t)c * Pu + lane] : 0.f;
  float lp = a.lp[c];

  // log pi and its gradient at the position whose element j is zq; every thread of the workgroup takes part
  auto evaluate = [&](float zq, float& lp_out, float& g_out) {
    __syncthreads();                       // previous readers of the LDS arrays are done
    if (wave == 0 && zl) sh_z[lane] = zq;
    __syncthreads();
    // theta_j = f(z)_j and this element's share of the forward log-det-Jacobian
    float th_j = 0.f, fl_j = 0.f;
    if (tl) {
      if (t_bij == BVI_BIJ_SOFTMAX_CENTERED) {
        float m = 0.f;
        for (int i = 0; i < t_size - 1; ++i) m = fmaxf(m, sh_z[t_zoff + i]);
        float s = expf(-m);
        for (int i = 0; i < t_size - 1; ++i) s += expf(sh_z[t_zoff + i] - m);
        const float ly = (t_j < t_size - 1 ? sh_z[t_zoff + t_j] : 0.f) - (m + logf(s));
        th_j = expf(ly);
        fl_j = ly + (t_j == 0 ? 0.5f * logf((float)t_size) : 0.f);
      } else {
        const float x = sh_z[t_zoff + t_j];
        if (t_bij == BVI_BIJ_SOFTPLUS) { th_j = softplusf(x); fl_j = -softplusf(-x); }
        else if (t_bij == BVI_BIJ_EXP) { th_j = expf(x); fl_j = x; }
        else th_j = x;
      }
    }
    if (wave == 0 && tl) sh_theta[lane] = th_j;
    __syncthreads();

    // ---- data loop: raw sums over all rows, private per thread ----
    float th[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) th[i] = 0.f;
    float acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = 0.f;
    if constexpr (FAM == BVI_FAMILY_NORMAL_IID) {
      const float mu = sh_theta[0];
#pragma unroll 4
      for (int64_t i = tid; i < a.n; i += nthreads) {
        const float r = a.y[i] - mu;
        acc[0] += r;
        acc[1] = fmaf(r, r, acc[1]);
      }
    } else if constexpr (FAM == BVI_FAMILY_LINREG) {
      // theta = (b0, w[d], scale); acc = (S2, S1, G[d])
      const float b0 = sh_theta[0];
#pragma unroll
      for (int j = 0; j < NP; ++j) th[j] = j < a.d ? sh_theta[1 + j] : 0.f;
#pragma unroll 4
      for (int64_t i = tid; i < a.n; i += nthreads) {
        float x[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) x[j] = j < a.d ? a.X[i * a.d + j] : 0.f;
        float r = a.y[i] - b0;
#pragma unroll
        for (int j = 0; j < NP; ++j) r = fmaf(-x[j], th[j], r);
        acc[0] = fmaf(r, r, acc[0]);
        acc[1] += r;
#pragma unroll
        for (int j = 0; j < NP; ++j) acc[2 + j] = fmaf(r, x[j], acc[2 + j]);
      }
    } else {
      // theta = (w[k], mu[k], s[k]); acc = (sum LSE, A0[k], A1[k], A2[k]) with u = (y - mu)/s
      float B[NP], mu[NP], isg[NP];
      float wsum = 0.f;
      for (int j = 0; j < a.k; ++j) wsum += sh_theta[j];
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        if (j < a.k) {
          const float sc = sh_theta[2 * a.k + j];
          mu[j] = sh_theta[a.k + j];
          isg[j] = 1.f / sc;
          B[j] = logf(sh_theta[j]) - logf(wsum) - logf(sc) - 0.5f * BVI_LN2PI;
        } else {
          mu[j] = 0.f; isg[j] = 0.f; B[j] = -INFINITY;
        }
      }
#pragma unroll 4
      for (int64_t i = tid; i < a.n; i += nthreads) {
        const float yv = a.y[i];
        float u[NP], e[NP], m = -INFINITY;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          u[j] = (yv - mu[j]) * isg[j];
          e[j] = fmaf(-0.5f * u[j], u[j], B[j]);
          m = fmaxf(m, e[j]);
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          e[j] = expf(e[j] - m);
          s += e[j];
        }
        acc[0] += m + logf(s);
        const float inv = 1.f / s;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          const float rho = e[j] * inv, ru = rho * u[j];
          acc[1 + j] += rho;
          acc[1 + NP + j] += ru;
          acc[1 + 2 * NP + j] = fmaf(ru, u[j], acc[1 + 2 * NP + j]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float v = wave_sum(acc[q]);
      if (lane == 0) sh_part[wave][q] = v;
    }
    __syncthreads();
    auto total = [&](int q) {
      float v = sh_part[0][q];
      for (int w = 1; w < nwaves; ++w) v += sh_part[w][q];
      return v;
    };

    // ---- lane j: d log lik / d theta_j, prior, chain rule ----
    float ll = 0.f, gth_j = 0.f;
    if constexpr (FAM == BVI_FAMILY_NORMAL_IID) {
      const float s1 = total(0), s2 = total(1), sc = sh_theta[1], inv2 = 1.f / (sc * sc);
      ll = -0.5f * s2 * inv2 - nf * logf(sc) - 0.5f * nf * BVI_LN2PI;
      gth_j = lane == 0 ? s1 * inv2 : (-nf / sc + s2 * inv2 / sc);
    } else if constexpr (FAM == BVI_FAMILY_LINREG) {
      const float s2 = total(0), sc = sh_theta[a.d + 1], inv2 = 1.f / (sc * sc);
      ll = -0.5f * s2 * inv2 - nf * logf(sc) - 0.5f * nf * BVI_LN2PI;
      if (lane == 0) gth_j = total(1) * inv2;
      else if (lane <= a.d) gth_j = total(1 + lane) * inv2;
      else gth_j = -nf / sc + s2 * inv2 / sc;
    } else {
      ll = total(0);
      if (tl) {
        const int part = lane / a.k, kk = lane - part * a.k;
        const float a0 = total(1 + kk), sc = sh_theta[2 * a.k + kk];
        if (part == 0) {
          float wsum = 0.f;
          for (int j = 0; j < a.k; ++j) wsum += sh_theta[j];
          gth_j = a0 / sh_theta[kk] - nf / wsum;
        } else if (part == 1) {
          gth_j = total(1 + NP + kk) / sc;
        } else {
          gth_j = (total(1 + 2 * NP + kk) - a0) / sc;
        }
      }
    }
    float lpr_j = 0.f;
    if (tl) {
      if (t_prior == BVI_PRIOR_NORMAL) {
        const float u = (th_j - t_a) / t_b;
        lpr_j = -0.5f * u * u - 0.5f * BVI_LN2PI - logf(t_b);
        gth_j += -u / t_b;
      } else if (t_prior == BVI_PRIOR_HALFNORMAL) {
        const float u = th_j / t_a;
        lpr_j = th_j < 0.f ? -INFINITY : (0.5f * logf(2.f / 3.14159265358979f) - logf(t_a) - 0.5f * u * u);
        gth_j += -u / t_a;
      } else if (t_prior == BVI_PRIOR_DIRICHLET) {
        lpr_j = (t_a - 1.f) * logf(th_j) + (t_j == 0 ? lgammaf(t_a * t_size) - t_size * lgammaf(t_a) : 0.f);
        gth_j += (t_a - 1.f) / th_j;
      }
    } else {
      gth_j = 0.f;
    }
    lp_out = ll + wave_sum(tl ? lpr_j + fl_j : 0.f);
    if (wave == 0 && tl) sh_gth[lane] = gth_j;
    __syncthreads();
    float gz = 0.f;
    if (zl) {
      if (z_bij == BVI_BIJ_SOFTMAX_CENTERED) {
        float inner = 0.f;
        for (int i = 0; i < z_size; ++i) inner += sh_theta[z_toff + i] * sh_gth[z_toff + i];
        const float t = sh_theta[z_toff + z_j];
        gz = t * (sh_gth[z_toff + z_j] - inner) + 1.f - (float)z_size * t;
      } else {
        const float gt = sh_gth[z_toff + z_j];
        if (z_bij == BVI_BIJ_SOFTPLUS) gz = gt * sigmoidf(zq) + sigmoidf(-zq);
        else if (z_bij == BVI_BIJ_EXP) gz = gt * expf(zq) + 1.f;
        else gz = gt;
      }
    }
    g_out = gz;
  };

  if (a.evaluate_initial) evaluate(zj, lp, gj);
  const float eps = a.eps_chain ? a.eps_chain[c] : a.eps;

  if constexpr (ALGO == 0) {
    for (int t = 0; t < a.T; ++t) {
      const uint64_t step = a.step0 + (uint64_t)t;
      float pj = 0.f;
      if (zl) {
        if (a.momenta_in) {
          pj = a.momenta_in[((int64_t)t * a.C + c) * Pu + lane];
        } else {
          float v[4];
          normal4(draw(a.seed, step, c, lane >> 2, 0), v);
          pj = v[lane & 3];
        }
      }
      const float z0 = zj, g0 = gj, lp0 = lp;
      const float k0 = wave_sum(0.5f * pj * pj);
      for (int l = 0; l < a.L; ++l) {
        pj = fmaf(0.5f * eps, gj, pj);
        zj = fmaf(eps, pj, zj);
        evaluate(zj, lp, gj);
        pj = fmaf(0.5f * eps, gj, pj);
      }
      const float k1 = wave_sum(0.5f * pj * pj);
      float lr = (lp - k1) - (lp0 - k0);
      if (!isfinite(lr)) lr = -INFINITY;
      const float lu = a.log_u_in ? a.log_u_in[(int64_t)t * a.C + c] : logf(u01(draw(a.seed, step, c, 0, 1).x));
      const bool acc = lu < lr;
      if (!acc) {
        zj = z0;
        gj = g0;
        lp = lp0;
      }
      if (wave == 0) {
        if (a.z_out && zl) a.z_out[((int64_t)t * a.C + c) * Pu + lane] = zj;
        if (lane == 0) {
          if (a.lp_out) a.lp_out[(int64_t)t * a.C + c] = lp;
          if (a.acc_out) a.acc_out[(int64_t)t * a.C + c] = acc ? 1 : 0;
          if (a.lr_out) a.lr_out[(int64_t)t * a.C + c] = lr;
        }
      }
    }
  } else if constexpr (ALGO == 2) {
    // random-walk Metropolis: z' = z + scale * N(0, I) (scale in a.eps), accept on the log-density ratio
    for (int t = 0; t < a.T; ++t) {
      const uint64_t step = a.step0 + (uint64_t)t;
      float zp = 0.f;
      if (zl) {
        float v[4];
        normal4(draw(a.seed, step, c, lane >> 2, 0), v);
        zp = fmaf(eps, v[lane & 3], zj);
      }
      float lpp, gp;
      evaluate(zp, lpp, gp);
      float lr = lpp - lp;
      if (!isfinite(lr)) lr = -INFINITY;
      const bool acc = logf(u01(draw(a.seed, step, c, 0, 1).x)) < lr;
      if (acc) {
        zj = zp;
        gj = gp;
        lp = lpp;
      }
      if (wave == 0) {
        if (a.z_out && zl) a.z_out[((int64_t)t * a.C + c) * Pu + lane] = zj;
        if (lane == 0) {
          if (a.lp_out) a.lp_out[(int64_t)t * a.C + c] = lp;
          if (a.acc_out) a.acc_out[(int64_t)t * a.C + c] = acc ? 1 : 0;
          if (a.lr_out) a.lr_out[(int64_t)t * a.C + c] = lr;
        }
      }
    }
  } else {
    // sum_j a_j b_j over the chain's elements (lanes beyond Pu hold zeros)
    auto turning = [&](float aq, float ap, float bq, float bp) {
      const float dq = bq - aq;
      return wave_sum(dq * ap) <= 0.f || wave_sum(dq * bp) <= 0.f;
    };
    for (int t = 0; t < a.T; ++t) {
      const uint64_t step = a.step0 + (uint64_t)t;
      float p0 = 0.f;
      if (zl) {
        float v[4];
        normal4(draw(a.seed, step, c, lane >> 2, 0), v);
        p0 = v[lane & 3];
      }
      const float h0 = -lp + wave_sum(0.5f * p0 * p0);
      float energy = -h0;       // log-density minus kinetic energy at the selected point (tfp's kernel_results.energy)
      float ql = zj, pl = p0, gl = gj, qr = zj, pr = p0, gr = gj;     // trajectory ends (left = backward in time)
      float log_w = 0.f, sum_acc = 0.f;
      int n_leap = 0, leaf_ctr = 0;
      bool active = true, diverged = false, moved = false;
      for (int depth = 0; depth < a.max_depth && active; ++depth) {
        const U4 rd = draw(a.seed, step, c, depth, 2);
        const bool go_right = u01(rd.x) < 0.5f;
        const float log_u_sub = logf(u01(rd.y));
        const float se = go_right ? eps : -eps;
        float q = go_right ? qr : ql, p = go_right ? pr : pl, g = go_right ? gr : gl;
        float sub_log_w = -INFINITY, sub_q = zj, sub_g = gj, sub_lp = lp, sub_e = energy;
        bool sub_ok = true, turn_sub = false;
        const int n_leaves = 1 << depth;
        for (int leaf = 0; leaf < n_leaves; ++leaf) {
          float lpq;
          p = fmaf(0.5f * se, g, p);
          q = fmaf(se, p, q);
          evaluate(q, lpq, g);
          p = fmaf(0.5f * se, g, p);
          float dh = (-lpq + wave_sum(0.5f * p * p)) - h0;
          if (!isfinite(dh)) dh = INFINITY;
          const bool div_now = dh > a.max_energy_diff;
          diverged |= div_now;
          const bool ok = !div_now;
          n_leap += 1;
          sum_acc += expf(fminf(-dh, 0.f));
          const float lw = -dh, new_w = log_add_exp(sub_log_w, lw);
          const float log_ul = logf(u01(draw(a.seed, step, c, leaf_ctr++, 3).x));
          if (ok && log_ul < lw - new_w) {       // multinomial sample within the subtree
            sub_q = q;
            sub_g = g;
            sub_lp = lpq;
            sub_e = -(dh + h0);
          }
          if (ok) sub_log_w = new_w;
          // U-turn checks inside the subtree against checkpoints of the left ends of its sub-subtrees:
          // slots popcount(leaf >> 1) - (trailing ones of leaf) + 1 .. popcount(leaf >> 1)
          const int hi = __popc((unsigned)leaf >> 1);
          if ((leaf & 1) == 0) {
            if (lane < kMaxP) {
              sh_ckq[hi][lane] = q;
              sh_ckp[hi][lane] = p;
            }
          } else {
            const int lo = hi - (__ffs(~leaf) - 1) + 1;
            for (int k = lo; k <= hi; ++k) {
              const float cq = sh_ckq[k][lane], cp = sh_ckp[k][lane];
              // ordered in time: for leftward subtrees the checkpoint is the later end
              const bool tn = go_right ? turning(cq, cp, q, p) : turning(q, p, cq, cp);
              turn_sub |= ok && tn;
            }
          }
          sub_ok = ok && !turn_sub;
          if (!sub_ok) break;
        }
        if (go_right) { qr = q; pr = p; gr = g; } else { ql = q; pl = p; gl = g; }
        const bool complete = sub_ok;
        if (complete && log_u_sub < sub_log_w - log_w) {     // biased progressive sampling between subtrees
          zj = sub_q;
          gj = sub_g;
          lp = sub_lp;
          energy = sub_e;
          moved = true;
        }
        if (complete) log_w = log_add_exp(log_w, sub_log_w);
        active = complete && !turning(ql, pl, qr, pr);
      }
      if (wave == 0) {
        if (a.z_out && zl) a.z_out[((int64_t)t * a.C + c) * Pu + lane] = zj;
        if (lane == 0) {
          const int64_t o = (int64_t)t * a.C + c;
          if (a.lp_out) a.lp_out[o] = lp;
          if (a.acc_out) a.acc_out[o] = moved ? 1 : 0;
          if (a.lr_out) a.lr_out[o] = logf(fmaxf(sum_acc / (float)(n_leap > 0 ? n_leap : 1), 1e-30f));
          if (a.leap_out) a.leap_out[o] = n_leap;
          if (a.div_out) a.div_out[o] = diverged ? 1 : 0;
          if (a.energy_out) a.energy_out[o] = energy;
        }
      }
    }
  }
  if (wave == 0) {
    if (zl) {
      a.z[(int64_t)c * Pu + lane] = zj;
      a.g[(int64_t)c * Pu + lane] = gj;
    }
    if (lane == 0) a.lp[c] = lp;
  }
}

template <int FAM, int NP>
void launch(const Blocks& bl, const FusedArgs& a, int algo, int threads, hipStream_t s) {
  if (algo == 0) hipLaunchKernelGGL((hmc_fused_kernel<FAM, NP, 0>), dim3(a.C), dim3(threads), 0, s, bl, a);
  else if (algo == 2) hipLaunchKernelGGL((hmc_fused_kernel<FAM, NP, 2>), dim3(a.C), dim3(threads), 0, s, bl, a);
  else hipLaunchKernelGGL((hmc_fused_kernel<FAM, NP, 1>), dim3(a.C), dim3(threads), 0, s, bl, a);
}

// common argument checks and dispatch of both entry points
int run_fused(int algo, int family, const float* X, const float* y, int64_t n, int d, int k,
              const bvi_param_block* blocks, int n_blocks, FusedArgs& a, void* stream) {
  Blocks bl;
  const int st = make_blocks(blocks, n_blocks, &bl, &a.P, &a.Pu);
  if (st != BVI_OK) return st;
  if ((!y && n > 0) || n < 0 || a.C < 1 || a.T < 0 || !a.z || !a.g || !a.lp) return BVI_ERR_ARG;
  int want_p;
  if (family == BVI_FAMILY_NORMAL_IID) want_p = 2;
  else if (family == BVI_FAMILY_LINREG) {
    if ((!X && n > 0) || d < 1) return BVI_ERR_ARG;
    if (d > 16) return BVI_ERR_UNSUPPORTED;
    want_p = d + 2;
  } else if (family == BVI_FAMILY_GMM1D) {
    if (k < 1) return BVI_ERR_ARG;
    if (k > 8) return BVI_ERR_UNSUPPORTED;
    want_p = 3 * k;
  } else {
    return BVI_ERR_UNSUPPORTED;
  }
  if (a.P != want_p || a.P > kMaxP) return BVI_ERR_ARG;
  a.X = X; a.y = y; a.n = n; a.d = d; a.k = k;
  if (a.T == 0 && !a.evaluate_initial) return BVI_OK;
  const int threads = n <= 1024 ? 64 : 256;
  hipStream_t s = (hipStream_t)stream;
  if (family == BVI_FAMILY_NORMAL_IID) launch<BVI_FAMILY_NORMAL_IID, 1>(bl, a, algo, threads, s);
  else if (family == BVI_FAMILY_LINREG) {
    if (d <= 1) launch<BVI_FAMILY_LINREG, 1>(bl, a, algo, threads, s);
    else if (d <= 2) launch<BVI_FAMILY_LINREG, 2>(bl, a, algo, threads, s);
    else if (d <= 4) launch<BVI_FAMILY_LINREG, 4>(bl, a, algo, threads, s);
    else if (d <= 8) launch<BVI_FAMILY_LINREG, 8>(bl, a, algo, threads, s);
    else launch<BVI_FAMILY_LINREG, 16>(bl, a, algo, threads, s);
  } else {
    if (k <= 2) launch<BVI_FAMILY_GMM1D, 2>(bl, a, algo, threads, s);
    else if (k <= 4) launch<BVI_FAMILY_GMM1D, 4>(bl, a, algo, threads, s);
    else launch<BVI_FAMILY_GMM1D, 8>(bl, a, algo, threads, s);
  }
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

}  // namespace

extern "C" int bvi_hmc_fused(int family, const float* X, const float* y, int64_t n, int d, int k,
                             const bvi_param_block* blocks, int n_blocks, int n_chains, float step_size,
                             const float* step_sizes, int n_leapfrog, int n_transitions, uint64_t seed,
                             uint64_t first_step, int evaluate_initial, float* z, float* grad_z, float* lp,
                             const float* momenta_in, const float* log_u_in, float* z_out, float* lp_out,
                             unsigned char* accepted_out, float* log_ratio_out, void* stream) {
  if (n_leapfrog < 1) return BVI_ERR_ARG;
  FusedArgs a = {};
  a.C = n_chains; a.T = n_transitions; a.L = n_leapfrog;
  a.eps = step_size; a.eps_chain = step_sizes; a.seed = seed; a.step0 = first_step;
  a.evaluate_initial = evaluate_initial; a.z = z; a.g = grad_z; a.lp = lp; a.momenta_in = momenta_in;
  a.log_u_in = log_u_in; a.z_out = z_out; a.lp_out = lp_out; a.acc_out = accepted_out; a.lr_out = log_ratio_out;
  return run_fused(0, family, X, y, n, d, k, blocks, n_blocks, a, stream);
}

extern "C" int bvi_rwm_fused(int family, const float* X, const float* y, int64_t n, int d, int k,
                             const bvi_param_block* blocks, int n_blocks, int n_chains, float proposal_scale,
                             int n_transitions, uint64_t seed, uint64_t first_step, int evaluate_initial, float* z,
                             float* grad_z, float* lp, float* z_out, float* lp_out, unsigned char* accepted_out,
                             float* log_ratio_out, void* stream) {
  if (!(proposal_scale > 0.f)) return BVI_ERR_ARG;
  FusedArgs a = {};
  a.C = n_chains; a.T = n_transitions; a.L = 1;
  a.eps = proposal_scale; a.seed = seed; a.step0 = first_step;
  a.evaluate_initial = evaluate_initial; a.z = z; a.g = grad_z; a.lp = lp;
  a.z_out = z_out; a.lp_out = lp_out; a.acc_out = accepted_out; a.lr_out = log_ratio_out;
  return run_fused(2, family, X, y, n, d, k, blocks, n_blocks, a, stream);
}

extern "C" int bvi_nuts_fused(int family, const float* X, const float* y, int64_t n, int d, int k,
                              const bvi_param_block* blocks, int n_blocks, int n_chains, float step_size,
                              const float* step_sizes, int max_tree_depth, float max_energy_diff,
                              int n_transitions, uint64_t seed, uint64_t first_step, int evaluate_initial, float* z,
                              float* grad_z, float* lp, float* z_out, float* lp_out, unsigned char* moved_out,
                              float* log_accept_out, int* leapfrogs_out, unsigned char* diverged_out,
                              float* energy_out, void* stream) {
  if (max_tree_depth < 1 || max_tree_depth > kMaxDepth || !(max_energy_diff > 0.f)) return BVI_ERR_ARG;
  FusedArgs a = {};
  a.C = n_chains; a.T = n_transitions; a.L = 1;
  a.eps = step_size; a.eps_chain = step_sizes; a.seed = seed; a.step0 = first_step;
  a.evaluate_initial = evaluate_initial; a.z = z; a.g = grad_z; a.lp = lp;
  a.z_out = z_out; a.lp_out = lp_out; a.acc_out = moved_out; a.lr_out = log_accept_out;
  a.max_depth = max_tree_depth; a.max_energy_diff = max_energy_diff; a.leap_out = leapfrogs_out;
  a.div_out = diverged_out; a.energy_out = energy_out;
  return run_fused(1, family, X, y, n, d, k, blocks, n_blocks, a, stream);
}
