// The No-U-Turn sampler for datasets that need the batched likelihood kernels (one fused pass over
// the data per leapfrog step, shared by all chains): every piece of per-chain tree bookkeeping of
// tfp.mcmc.NoUTurnSampler (bayes_vi/inference/mcmc/transition_kernels.py:238-246) -- direction and
// subtree set-up, energy and divergence test, multinomial sampling inside the subtree, checkpointed
// U-turn checks, biased progressive sampling between subtrees, the stopping rule -- runs on the device,
// one workgroup per chain (64 to 1024 threads, by the size of the parameter vector), threads over its
// elements.  The host only sequences the launches
//   begin; per depth { subtree_begin; per leaf { leapfrog_pre_chainwise, <family>_logp_grad
//   [, all-reduce], leapfrog_post_chainwise, leaf }; subtree_end }
// and reads one word per second leaf / per depth to learn whether any chain is still building / active.
// Chains advance in lock-step; a chain that is done gets step size 0 (bvi_nuts_state.se), which freezes
// it, and is ignored by the bookkeeping.  Same algorithm and the same Philox draws as the
// one-workgroup-per-chain kernel in hmc_fused.hip (momenta: stream 0; direction / subtree-accept
// uniform: words 0, 1 of block `depth`, stream 2; multinomial uniform of the n-th leapfrog: word 0 of
// block n, stream 3), so both follow the fp64 restatement leaf by leaf.
#include "bvi_common.h"
#include "posterior_device.h"

namespace {

enum : int { F_ACTIVE = 1, F_SUB_OK = 2, F_TURN = 4, F_DIVERGED = 8, F_MOVED = 16, F_RIGHT = 32 };

__device__ __forceinline__ float nb_log_add_exp(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + logf(expf(a - m) + expf(b - m));
}

// (b_q - a_q) . a_p <= 0 or (b_q - a_q) . b_p <= 0, vectors of one chain, all threads of the workgroup take part
__device__ __forceinline__ bool nb_turning(const float* aq, const float* ap, const float* bq, const float* bp, int Pu,
                                           float* red) {
  float d1 = 0.f, d2 = 0.f;
  for (int j = threadIdx.x; j < Pu; j += blockDim.x) {
    const float dq = bq[j] - aq[j];
    d1 = fmaf(dq, ap[j], d1);
    d2 = fmaf(dq, bp[j], d2);
  }
  d1 = block_reduce<false>(d1, red);
  d2 = block_reduce<false>(d2, red);
  return d1 <= 0.f || d2 <= 0.f;
}

__global__ __launch_bounds__(1024) void nuts_begin_kernel(bvi_nuts_state s, uint64_t seed, uint64_t step) {
  __shared__ float red[16];
  const int c = blockIdx.x, lane = threadIdx.x, nt = blockDim.x, Pu = s.n_unconstrained;
  const int64_t o = (int64_t)c * Pu;
  float ke = 0.f;
  for (int j = lane; j < Pu; j += nt) {
    float v[4];
    normal4(draw(seed, step, c, j >> 2, 0), v);
    const float p0 = v[j & 3];
    ke = fmaf(0.5f * p0, p0, ke);
    s.pl[o + j] = p0; s.pr[o + j] = p0;
    s.ql[o + j] = s.z[o + j]; s.qr[o + j] = s.z[o + j];
    s.gl[o + j] = s.grad[o + j]; s.gr[o + j] = s.grad[o + j];
  }
  ke = block_reduce<false>(ke, red);
  if (lane == 0) {
    s.h0[c] = -s.lp[c] + ke;
    s.energy[c] = s.lp[c] - ke;           // of the selected point: the initial one until a subtree replaces it
    s.log_w[c] = 0.f;
    s.sum_acc[c] = 0.f;
    s.n_leap[c] = 0;
    s.leaf_ctr[c] = 0;
    s.flags[c] = F_ACTIVE;
  }
}

__global__ __launch_bounds__(1024) void nuts_subtree_begin_kernel(bvi_nuts_state s, uint64_t seed, uint64_t step,
                                                                int depth) {
  const int c = blockIdx.x, lane = threadIdx.x, nt = blockDim.x, Pu = s.n_unconstrained;
  const int64_t o = (int64_t)c * Pu;
  const int fl = s.flags[c];
  const bool active = fl & F_ACTIVE;
  const U4 rd = draw(seed, step, c, depth, 2);
  const bool go_right = u01(rd.x) < 0.5f;
  if (active) {
    for (int j = lane; j < Pu; j += nt) {
      s.q[o + j] = go_right ? s.qr[o + j] : s.ql[o + j];
      s.p[o + j] = go_right ? s.pr[o + j] : s.pl[o + j];
      s.gq[o + j] = go_right ? s.gr[o + j] : s.gl[o + j];
      s.sub_q[o + j] = s.z[o + j];
      s.sub_g[o + j] = s.grad[o + j];
    }
  }
  __syncthreads();                          // every thread has read the chain's scalars
  if (lane == 0) {
    s.se[c] = active ? (go_right ? s.step_size : -s.step_size) : 0.f;
    s.log_u_sub[c] = logf(u01(rd.y));
    s.sub_log_w[c] = -INFINITY;
    s.sub_lp[c] = s.lp[c];
    s.sub_energy[c] = s.energy[c];
    s.flags[c] = (fl & ~(F_SUB_OK | F_TURN | F_RIGHT)) | (active ? F_SUB_OK : 0) | (go_right ? F_RIGHT : 0);
  }
}

// after leapfrog_post_chainwise: the cursor (q, p, gq, lpq) is one leapfrog step further for every chain
// that is still building its subtree
__global__ __launch_bounds__(1024) void nuts_leaf_kernel(bvi_nuts_state s, uint64_t seed, uint64_t step, int leaf,
                                                       int* __restrict__ any_building) {
  __shared__ float red[16];
  const int c = blockIdx.x, lane = threadIdx.x, nt = blockDim.x, Pu = s.n_unconstrained, C = s.n_chains;
  const int64_t o = (int64_t)c * Pu;
  int fl = s.flags[c];
  if (!(fl & F_SUB_OK)) return;
  const bool go_right = fl & F_RIGHT;
  float ke = 0.f;
  for (int j = lane; j < Pu; j += nt) ke = fmaf(0.5f * s.p[o + j], s.p[o + j], ke);
  ke = block_reduce<false>(ke, red);
  const float lpq = s.lpq[c];
  float dh = (float)(((double)ke - (double)lpq) - (double)s.h0[c]);   // small differences of stored levels: in double
  if (!isfinite(dh)) dh = INFINITY;
  const bool div_now = dh > s.max_energy_diff;
  const bool ok = !div_now;
  const float lw = -dh, sub_w = s.sub_log_w[c], new_w = nb_log_add_exp(sub_w, lw);
  const int ctr = s.leaf_ctr[c];
  const float log_ul = logf(u01(draw(seed, step, c, ctr, 3).x));
  if (ok && log_ul < lw - new_w) {          // multinomial sample within the subtree
    for (int j = lane; j < Pu; j += nt) {
      s.sub_q[o + j] = s.q[o + j];
      s.sub_g[o + j] = s.gq[o + j];
    }
    if (lane == 0) {
      s.sub_lp[c] = lpq;
      s.sub_energy[c] = lpq - ke;
    }
  }
  // U-turn checks against the checkpoints of the left ends of the sub-subtrees this leaf completes
  bool turn = fl & F_TURN;
  const int hi = __popc((unsigned)leaf >> 1);
  if ((leaf & 1) == 0) {
    float* cq = s.ck_q + ((int64_t)hi * C + c) * Pu;
    float* cp = s.ck_p + ((int64_t)hi * C + c) * Pu;
    for (int j = lane; j < Pu; j += nt) { cq[j] = s.q[o + j]; cp[j] = s.p[o + j]; }
  } else {
    const int lo = hi - (__ffs(~leaf) - 1) + 1;
    for (int k = lo; k <= hi; ++k) {
      const float* cq = s.ck_q + ((int64_t)k * C + c) * Pu;
      const float* cp = s.ck_p + ((int64_t)k * C + c) * Pu;
      const bool tn = go_right ? nb_turning(cq, cp, s.q + o, s.p + o, Pu, red)
                               : nb_turning(s.q + o, s.p + o, cq, cp, Pu, red);
      turn |= ok && tn;
    }
  }
  const bool sub_ok = ok && !turn;
  __syncthreads();                          // every thread has read the chain's scalars
  if (lane == 0) {
    s.n_leap[c] += 1;
    s.leaf_ctr[c] = ctr + 1;
    s.sum_acc[c] += expf(fminf(-dh, 0.f));
    if (ok) s.sub_log_w[c] = new_w;
    fl = (fl & ~(F_SUB_OK | F_TURN)) | (sub_ok ? F_SUB_OK : 0) | (turn ? F_TURN : 0) | (div_now ? F_DIVERGED : 0);
    s.flags[c] = fl;
    if (sub_ok) atomicOr(any_building, 1);
    else s.se[c] = 0.f;                     // freezes this chain's cursor for the rest of the subtree
  }
}

__global__ __launch_bounds__(1024) void nuts_subtree_end_kernel(bvi_nuts_state s, int* __restrict__ any_active) {
  __shared__ float red[16];
  const int c = blockIdx.x, lane = threadIdx.x, nt = blockDim.x, Pu = s.n_unconstrained;
  const int64_t o = (int64_t)c * Pu;
  int fl = s.flags[c];
  if (!(fl & F_ACTIVE)) return;
  const bool go_right = fl & F_RIGHT;
  for (int j = lane; j < Pu; j += nt) {     // the trajectory end moves to where the subtree stopped
    if (go_right) { s.qr[o + j] = s.q[o + j]; s.pr[o + j] = s.p[o + j]; s.gr[o + j] = s.gq[o + j]; }
    else { s.ql[o + j] = s.q[o + j]; s.pl[o + j] = s.p[o + j]; s.gl[o + j] = s.gq[o + j]; }
  }
  const bool complete = fl & F_SUB_OK;
  const float log_w = s.log_w[c], sub_w = s.sub_log_w[c];
  const bool accept = complete && s.log_u_sub[c] < sub_w - log_w;     // biased progressive sampling
  if (accept) {
    for (int j = lane; j < Pu; j += nt) { s.z[o + j] = s.sub_q[o + j]; s.grad[o + j] = s.sub_g[o + j]; }
  }
  __syncthreads();                          // the end points written above are read back by other lanes below
  const bool turning = nb_turning(s.ql + o, s.pl + o, s.qr + o, s.pr + o, Pu, red);
  const bool active = complete && !turning;
  if (lane == 0) {
    if (accept) {
      s.lp[c] = s.sub_lp[c];
      s.energy[c] = s.sub_energy[c];
    }
    if (complete) s.log_w[c] = nb_log_add_exp(log_w, sub_w);
    fl = (fl & ~(F_ACTIVE | F_SUB_OK)) | (active ? F_ACTIVE : 0) | (accept ? F_MOVED : 0);
    s.flags[c] = fl;
    s.se[c] = 0.f;
    if (active) atomicOr(any_active, 1);
  }
}

dim3 chain_block(const bvi_nuts_state* s) { return dim3(chain_threads(s->n_unconstrained)); }

int check_state(const bvi_nuts_state* s) {
  if (!s || s->n_chains < 1 || s->n_unconstrained < 1 || s->max_tree_depth < 1 || s->max_tree_depth > 10 ||
      !(s->max_energy_diff > 0.f))
    return BVI_ERR_ARG;
  const void* ptrs[] = {s->z, s->lp, s->grad, s->q, s->p, s->gq, s->lpq, s->ql, s->pl, s->gl, s->qr, s->pr, s->gr,
                        s->sub_q, s->sub_g, s->sub_lp, s->ck_q, s->ck_p, s->h0, s->log_w, s->sub_log_w, s->sum_acc,
                        s->log_u_sub, s->se, s->energy, s->sub_energy, s->n_leap, s->leaf_ctr, s->flags};
  for (const void* p : ptrs)
    if (!p) return BVI_ERR_ARG;
  return BVI_OK;
}

}  // namespace

extern "C" {

int bvi_nuts_begin(const bvi_nuts_state* state, uint64_t seed, uint64_t step, void* stream) {
  const int st = check_state(state);
  if (st != BVI_OK) return st;
  hipLaunchKernelGGL(nuts_begin_kernel, dim3(state->n_chains), chain_block(state), 0, (hipStream_t)stream, *state,
                     seed, step);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_nuts_subtree_begin(const bvi_nuts_state* state, uint64_t seed, uint64_t step, int depth, void* stream) {
  const int st = check_state(state);
  if (st != BVI_OK) return st;
  if (depth < 0 || depth >= state->max_tree_depth) return BVI_ERR_ARG;
  hipLaunchKernelGGL(nuts_subtree_begin_kernel, dim3(state->n_chains), chain_block(state), 0, (hipStream_t)stream,
                     *state,
                     seed, step, depth);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_nuts_leaf(const bvi_nuts_state* state, uint64_t seed, uint64_t step, int leaf, int* any_building,
                  void* stream) {
  const int st = check_state(state);
  if (st != BVI_OK) return st;
  if (leaf < 0 || !any_building) return BVI_ERR_ARG;
  hipLaunchKernelGGL(nuts_leaf_kernel, dim3(state->n_chains), chain_block(state), 0, (hipStream_t)stream, *state,
                     seed, step, leaf, any_building);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

int bvi_nuts_subtree_end(const bvi_nuts_state* state, int* any_active, void* stream) {
  const int st = check_state(state);
  if (st != BVI_OK) return st;
  if (!any_active) return BVI_ERR_ARG;
  hipLaunchKernelGGL(nuts_subtree_end_kernel, dim3(state->n_chains), chain_block(state), 0, (hipStream_t)stream,
                     *state,
                     any_active);
  BVI_HIP_CHECK_LAUNCH();
  return BVI_OK;
}

}  // extern "C"
