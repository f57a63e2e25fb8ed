/*
 * bvi.h -- C ABI of libbvi_hip.so: the MI355X (gfx950) implementation of the
 * bayes_vi data-parallel hot path, the batched log-likelihood and its gradient
 * for all chains at once.
 *
 * The reference (bayes_vi 0.3.0) has no FFI: its boundary is the Python closure
 *     target_log_prob(state) = joint.log_prob(**params, y=targets)
 * built at bayes_vi/inference/mcmc/mcmc.py:57-62 and bayes_vi/inference/vi/vi.py:46-52
 * and evaluated (value and autodiff gradient) by TensorFlow-Probability inside
 * every HMC/NUTS leapfrog step and every Monte-Carlo ELBO gradient.  Each entry
 * point below replaces the data-dependent part of that closure for one
 * likelihood family; the citation on each names the reference construct it
 * stands in for.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - nothing here allocates, frees or synchronises: scratch memory comes in
 *     through `ws` (size from bvi_workspace_bytes) and results are ready when
 *     `stream` reaches the end of the enqueued work;
 *   - return value: 0 = success, > 0 = a hipError_t, < 0 = BVI_ERR_*;
 *   - outputs are LIKELIHOOD-ONLY sums over the rows handed in:
 *         logp[c]    = sum_{i < n} log p(y_i | theta_c, x_i)
 *         grad[c, :] = d logp[c] / d theta_c          (constrained space)
 *     so partial results of row shards add up (one all-reduce across GPUs);
 *     priors and bijector Jacobians are added once by the caller;
 *   - `theta` is [n_chains, P] row-major float32 in the flat parameter order of
 *     the reference (param_names order, each part row-major flattened:
 *     bayes_vi/inference/mcmc/mcmc.py:83-99, bayes_vi/model/model.py:123-153);
 *   - `grad` may be NULL: value only (RandomWalkMetropolis,
 *     bayes_vi/inference/mcmc/transition_kernels.py:319-321);
 *   - `logp_ref` (device pointer to one double, or NULL for 0): a level that is subtracted from every
 *     logp[c] in double before it is rounded to float.  A float holds a log-likelihood of 1e7 (10M rows)
 *     to +-0.5, which is the size of the differences a Metropolis test decides on; callers that only use
 *     differences (every sampler) keep a level near the chains' current values there -- bvi_logp_rebase
 *     maintains one -- and the outputs stay small and exact.  The reference has no such argument: TFP
 *     carries float32 log-probs and takes the rounding.  Row shards each subtract their own level.
 *   - non-finite parameters give non-finite outputs, never an error: the
 *     sampler turns them into rejections as the reference does.
 */
#ifndef BVI_H
#define BVI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BVI_VERSION 112

#define BVI_OK 0
#define BVI_ERR_ARG (-1)          /* NULL pointer, negative size, bad enum        */
#define BVI_ERR_UNSUPPORTED (-2)  /* shape outside what the kernels cover         */
#define BVI_ERR_WORKSPACE (-3)    /* ws is NULL or smaller than bvi_workspace_bytes */
#define BVI_ERR_TOO_LARGE (-4)    /* beyond the fixed scratch of an allocation-free entry point */

/* likelihood families */
#define BVI_FAMILY_NORMAL_IID 0   /* tfd.Sample(tfd.Normal(loc, scale), N)              README.md:114-120 */
#define BVI_FAMILY_LINREG 1       /* tfd.Normal(beta_0 + X w, scale)                    README.md:747-756 */
#define BVI_FAMILY_GMM1D 2        /* MixtureSameFamily(Categorical(probs), Normal)      univariate-gaussian-mixture-test.ipynb cell 15 */
#define BVI_FAMILY_GMM_FULLCOV 3  /* MixtureSameFamily(Categorical, MultivariateNormalTriL)  ...density-estimation-test.ipynb cell 7 */
#define BVI_FAMILY_MVN_TRIL 4     /* MultivariateNormalTriL(loc, scale_tril)            multivariate-gaussian-test.ipynb cell 15 */

/* what the matrix parameter of BVI_FAMILY_MVN_TRIL is */
#define BVI_MVN_SCALE_TRIL 0            /* the scale S itself (lower triangle read)                               */
#define BVI_MVN_CHOL_PRECISION_SOLVE 1  /* a Cholesky factor P of the precision; S = tril((P P^T)^-1): what
                                           scale_tril = tf.linalg.cholesky_solve(P, eye) evaluates                 */

/* arithmetic of the chains x features x rows contraction (linear regression) */
#define BVI_PREC_AUTO 0           /* bvi_linreg_logp_grad (unpacked f32 X): BF16X6 when n_chains >= 128,
                                     n >= 16384 and d is 32 or 64, else F32.  bvi_linreg_packed_logp_grad:
                                     the arithmetic of the packed format (BVI_PACK_F16: f16 plane + exact
                                     closed-form remainders, the fastest path from ~16 chains on) */
#define BVI_PREC_F32 1            /* exact f32 (f32-input MFMA / VALU fma chains)                      */
#define BVI_PREC_BF16X3 2         /* bf16 2-piece split, 3 MFMA products per GEMM (~2^-18 per product) */
#define BVI_PREC_BF16X6 3         /* bf16 3-piece split (6 products, f32 grade) for the residual GEMM,
                                     2-piece (3 products) for the gradient GEMM                        */

/* ---- posterior assembly and HMC state updates on the device ---- */
#define BVI_MAX_PARAM_BLOCKS 16

/* constraining bijector of a parameter block, forward = constrain (bayes_vi/model/model.py:51-55, 112-116) */
#define BVI_BIJ_IDENTITY 0
#define BVI_BIJ_SOFTPLUS 1          /* positive parameters (HalfNormal prior)             */
#define BVI_BIJ_EXP 2
#define BVI_BIJ_SOFTMAX_CENTERED 3  /* R^{K-1} -> K-simplex (Dirichlet prior)             */
#define BVI_BIJ_FILL_SCALE_TRIL 4   /* size/(dim*dim) packed vectors [dim(dim+1)/2] -> lower-triangular
                                       [dim,dim] matrices, softplus(.) + shift on the diagonal
                                       (tfb.FillScaleTriL: the Cholesky factors of a mixture)      */

/* prior of a parameter block (README.md:114-117, 747-751; mixture notebook cell 15) */
#define BVI_PRIOR_NONE 0
#define BVI_PRIOR_NORMAL 1          /* Normal(loc = a, scale = b) on every element        */
#define BVI_PRIOR_HALFNORMAL 2      /* HalfNormal(scale = a) on every element             */
#define BVI_PRIOR_DIRICHLET 3       /* Dirichlet(concentration = a on all `size` entries) */
#define BVI_PRIOR_WISHART_TRIL 4    /* WishartTriL(df = a, scale_tril = b * I, input_output_cholesky=True) on every
                                       [dim, dim] Cholesky factor of a FILL_SCALE_TRIL block
                                       (multivariate-gaussian-test.ipynb cell 15)                          */

/* One entry of Model.priors / Model.constraining_bijectors (bayes_vi/model/model.py:106-116),
 * in param_names order.  `size` is the flat constrained size of the block. */
typedef struct bvi_param_block {
  int32_t bijector;
  int32_t size;
  int32_t prior;
  float a;
  float b;
  int32_t dim;   /* FILL_SCALE_TRIL: matrix dimension; 0 otherwise */
  float shift;   /* FILL_SCALE_TRIL: added to the softplus of the diagonal */
} bvi_param_block;

int bvi_version(void);

/* Human-readable text for a status returned by any entry point (host pointer,
 * static storage). */
const char* bvi_status_string(int status);

/* Bytes of device scratch the entry point of `family` needs for this shape.
 * d = features (LINREG) or dimensions (GMM_FULLCOV), k = mixture components;
 * pass 0 for the ones the family does not use. */
int bvi_workspace_bytes(int family, int64_t n_rows, int n_chains, int d, int k, size_t* out_bytes);

/* iid Normal likelihood: y[n], theta[C,2] = (loc, scale) -> logp[C], grad[C,2].
 * Replaces JointDistributionNamedAutoBatched.log_prob of the 'y' node built by
 * bayes_vi/model/model.py:17-21,197-203 for README.md:119-120, and its gradient
 * taken by tfp.mcmc kernels (transition_kernels.py:122-128, 238-246). */
int bvi_normal_iid_logp_grad(const float* y, int64_t n, const float* theta, int n_chains,
                             float* logp, float* grad, const double* logp_ref, void* ws, size_t ws_bytes,
                             void* stream);

/* The same computation as bvi_normal_iid_logp_grad WITHOUT a workspace argument and without any allocation:
 * the per-row-block partial sums live in a device array inside the library (2 MiB, held on every GPU the library
 * has run on until the process ends).  No counterpart in the reference: it exists for callers that cannot
 * allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * At most BVI_NOALLOC_MAX_CHAINS chains (any n); more returns BVI_ERR_TOO_LARGE.  Results are identical to
 * bvi_normal_iid_logp_grad's.  All calls on one GPU share that GPU's array: calls that overlap in time on one
 * GPU (different streams or threads) race; calls on one stream, or on different GPUs, do not.  The first call on
 * each GPU looks the array's address up (hipGetSymbolAddress): make it outside graph capture. */
#define BVI_NOALLOC_MAX_CHAINS 256
int bvi_normal_iid_logp_grad_noalloc(const float* y, int64_t n, const float* theta, int n_chains,
                                     float* logp, float* grad, void* stream);

/* Gaussian linear regression: X[n,d] row-major, y[n],
 * theta[C,d+2] = (beta_0, w_1..w_d, scale) -> logp[C], grad[C,d+2].
 * Replaces the feature-conditioned likelihood node of bayes_vi/model/model.py:194-195,
 * 201-203 for README.md:754-756 (features as one [N,D] tensor, utils/datasets.py:48-50).
 * This entry point reads the caller's f32 matrix as it is and keeps nothing between calls.  A caller that
 * evaluates the same dataset many times (every sampler does) should pack it once with
 * bvi_linreg_pack(BVI_PACK_F16) and call bvi_linreg_packed_logp_grad: at n = 10M, d = 64, 1024 chains that is
 * 3.0 ms per evaluation against 8.2 ms here, and half the HBM traffic at few chains; the pack step costs ~8 ms
 * there.  bvi_linreg_recommended_format says, for a shape, whether packing is the faster route. */
int bvi_linreg_logp_grad(const float* X, const float* y, int64_t n, int d, const float* theta,
                         int n_chains, int precision, float* logp, float* grad, const double* logp_ref,
                         void* ws, size_t ws_bytes, void* stream);

/* The same likelihood on a dataset packed once for the matrix-core kernels.
 * The reference turns the dataset into two immutable full-batch tensors when the
 * driver is constructed (bayes_vi/inference/mcmc/mcmc.py:52, vi/vi.py:40); packing is the
 * MI355X form of that one-time step, so that the per-evaluation kernel stages tiles by plain
 * copies.  Formats:
 *   BVI_PACK_BF16X3  d = 32 or 64; three bf16 piece planes [3][n_pad][d] + a 16-B row extension;
 *                    evaluated with precision BF16X6 (default) or BF16X3;
 *   BVI_PACK_F16     1 <= d <= 64; the columns are centred (x~ = x - shift, shift = the column means in f32;
 *                    the model's intercept becomes the fitted value at the centroid, b0 + shift . w, which is
 *                    orthogonal to every sum over the rows); a 1280-B header with per-column power-of-two scales
 *                    of x~, the shifts, data sums and max|y - mean y|, then [64][64] doubles X^T X,
 *                    A = X^T X - XH^T XH and XR^T XH (transposed), and 64 doubles each XR^T y and XR^T 1 (X: the
 *                    centred matrix, XH: its f16 plane, XR = X - XH, exact), then one f16 plane
 *                    [n_pad][32 or 64] of the scaled x~, zero beyond column d, + y[n_pad] in f32;
 *                    n_pad = n rounded up to 128; precision must be AUTO.  Both contractions over the rows run
 *                    on the f16 matrix cores (f16 plane times one f16 piece of the weights; residuals in 2
 *                    pieces times the f16 plane); what the f16 operands leave of X and of the weights, and
 *                    (for many chains) the intercept, enter every sum linearly and are added exactly, in
 *                    double, from those pack-time sums after the reduction.  Log-density at f32 accuracy,
 *                    gradient error at the level of the exact-f32 kernel's, for residuals down to ~1/5000 of
 *                    the fitted values and columns whose mean is up to ~1000 times their spread.
 *                    The format is meant for datasets of many tiles (the host layer packs from 16384 rows on): the
 *                    padding rows of the last tile take part with the residual of a reference chain and are taken
 *                    out again after the reduction, which costs a chain whose intercept lies far from its block's
 *                    first chain up to 128 / n of the float precision of its (then very large) log-density -- 2e-5
 *                    relative at n = 1, below 1e-7 from a few thousand rows on.
 * Padding rows are zero (BF16X3: n_pad = n rounded up to 64).  bvi_linreg_pack needs bvi_linreg_pack_scratch_bytes
 * of workspace (0 for BF16X3). */
#define BVI_PACK_BF16X3 0
#define BVI_PACK_F16 1
int bvi_linreg_packed_bytes(int64_t n, int d, int format, size_t* out_bytes);
/* Which format a caller should pack a (shard of a) dataset into for evaluations of `n_chains` chains under
 * BVI_PREC_AUTO: *format = BVI_PACK_F16 when the f16-plane kernel is the fastest route (more than 16 chains -- 32 for
 * d <= 8 -- on at least 16384 rows, or any number of chains from 400 000 rows of more than 8 features on), or -1
 * when bvi_linreg_logp_grad on the unpacked rows is (few chains on small shards; d > 64 is unsupported either
 * way).  bvi_linreg_logp_grad itself never packs (it allocates nothing and the pack step is one pass over the
 * data per dataset, not per call): with BVI_PREC_AUTO on unpacked rows and more than 16 chains it runs the
 * slower bf16-piece kernel (n = 10M, d = 64, 1024 chains: 8.2 ms against 3.0 ms), so a caller that evaluates a
 * dataset more than a few times asks here first.  No counterpart in the reference (TFP has one code path). */
int bvi_linreg_recommended_format(int64_t n, int d, int n_chains, int* format);
int bvi_linreg_pack_scratch_bytes(int d, int format, size_t* out_bytes);
int bvi_linreg_pack(const float* X, const float* y, int64_t n, int d, int format, void* packed,
                    size_t packed_bytes, void* ws, size_t ws_bytes, void* stream);
int bvi_linreg_packed_logp_grad(const void* packed, int64_t n, int d, int format, const float* theta,
                                int n_chains, int precision, float* logp, float* grad, const double* logp_ref,
                                void* ws, size_t ws_bytes, void* stream);

/* Univariate Gaussian mixture: y[n], theta[C,3k] = (weights_k, loc_k, scale_k)
 * -> logp[C], grad[C,3k].  Weights are normalised by their sum, as
 * Categorical(probs=weights) does.  Replaces the likelihood node for
 * exploratory_experiments/univariate-gaussian-mixture-test.ipynb cell 15.
 * The rows may come in any order.  A caller that keeps them in ascending order lets launches of up to 16 chains
 * leave out the components that are out of reach of a whole group of neighbouring rows (the shipped host layer
 * sorts its copy once; the sums then differ from those over another order in the last float digits). */
int bvi_gmm1d_logp_grad(const float* y, int64_t n, int k, const float* theta, int n_chains,
                        float* logp, float* grad, const double* logp_ref, void* ws, size_t ws_bytes, void* stream);

/* Full-covariance Gaussian mixture: Y[n,d] row-major,
 * theta[C, k + k*d + k*d*d] = (weights_k, locs[k][d], scale_tril[k][d][d] row-major, entries above the
 * diagonal ignored) -> logp[C], grad[C, same layout] (gradient of scale_tril is lower triangular).
 * d <= 32, k <= 16; f16-piece arithmetic on the matrix cores (data in 2 pieces, centred and range-normalised
 * per column by a statistics pass per call; whitening matrices in 3 pieces: exact), f32 accumulation.
 * The rows may come in any order.  Per call the library bounds, for every 64-row tile (centre and radius of its rows)
 * and every chain, each component's log-density over the tile (spectral-norm bound of the whitening matrix) and does
 * not evaluate a component on a tile where its responsibility is below 2^-46 for every row; tiles only one component
 * reaches are summed without a log-sum-exp, tiles two or three reach once per component by a wave of its own.  A caller that stores neighbouring rows next to each other (the shipped
 * host layer orders its copy cluster by cluster) gets most tiles of that kind: N = 1M, D = 32, K = 16, 64 chains
 * 1.5 ms against 13.7 ms for rows in no particular order.  What is left out is below 2^-40 of any sum.
 * Replaces the likelihood node
 * MixtureSameFamily(Categorical(probs), MultivariateNormalTriL(locs, scale_tril)) of
 * exploratory_experiments/multivariate-gaussian-mixture-density-estimation-test.ipynb cell 7. */
int bvi_gmm_fullcov_logp_grad(const float* Y, int64_t n, int d, int k, const float* theta, int n_chains,
                              float* logp, float* grad, const double* logp_ref, void* ws, size_t ws_bytes,
                              void* stream);

/* The part of bvi_gmm_fullcov_logp_grad / bvi_mvn_tril_logp_grad that depends on the rows alone, taken once per
 * dataset: column centres and scales, the centre and radius of every 64-row tile and of every group of 64 tiles, and
 * the rows' two f16 pieces (128 B per row).  The *_packed_logp_grad entry points take the rows AND the packed buffer
 * (the kernel of the tiles several components reach reads the rows themselves) and skip four launches and one pass over
 * the rows per call; results are bit-identical to the unpacked entry points'.  N = 125 000, D = 32, K = 16, 16 chains:
 * 0.34 against 0.37 ms per leapfrog step of the device NUTS.  packed must be 16-byte aligned.  No counterpart in the reference (TFP keeps
 * nothing between log_prob calls); the linear-regression analogue is bvi_linreg_pack. */
int bvi_gmm_fullcov_packed_bytes(int64_t n, int d, size_t* out_bytes);
int bvi_gmm_fullcov_pack_scratch_bytes(int d, size_t* out_bytes);
int bvi_gmm_fullcov_pack(const float* Y, int64_t n, int d, void* packed, size_t packed_bytes, void* ws,
                         size_t ws_bytes, void* stream);
int bvi_gmm_fullcov_packed_logp_grad(const float* Y, const void* packed, int64_t n, int d, int k, const float* theta,
                                     int n_chains, float* logp, float* grad, const double* logp_ref, void* ws,
                                     size_t ws_bytes, void* stream);
int bvi_mvn_tril_packed_logp_grad(const float* Y, const void* packed, int64_t n, int d, int parameterisation,
                                  const float* theta, int n_chains, float* logp, float* grad, const double* logp_ref,
                                  void* ws, size_t ws_bytes, void* stream);

/* One-time row order for bvi_gmm_fullcov_logp_grad: Y_out[n,d] = the rows of Y[n,d] stored so that rows lying close
 * together are neighbours (Lloyd's algorithm on n_centres <= BVI_CLUSTER_MAX_CENTRES centres started from rows at a
 * fixed stride, two updates, then a stable counting sort by nearest centre; a few n_centres-by-d distance passes over
 * the rows).  order (optional, int64[n]): order[i] = the row of Y that Y_out's row i is.  The result is the same bit
 * for bit on every run.  d <= 32; Y_out must not alias Y; rows with a non-finite entry are kept (with centre 0).
 * 4 k centres (at least 16) is what the shipped host layer asks for a k-component mixture.
 * No counterpart in the reference: the rows of a likelihood sum are exchangeable (the reference reduces them with
 * tf.reduce_sum, bayes_vi/model/model.py:17-21), so the order in memory is this library's to choose. */
#define BVI_CLUSTER_MAX_CENTRES 128
int bvi_cluster_rows_scratch_bytes(int64_t n, int d, int n_centres, size_t* out_bytes);
int bvi_cluster_rows(const float* Y, int64_t n, int d, int n_centres, float* Y_out, int64_t* order, void* ws,
                     size_t ws_bytes, void* stream);

/* Single multivariate normal: Y[n,d] row-major, theta[C, d + d*d] = (loc[d], M[d][d] row-major, entries above the
 * diagonal ignored) -> logp[C], grad[C, same layout] (gradient of M lower triangular).  `parameterisation` says
 * what M is (BVI_MVN_*).  d <= 32.  The sum over the rows is the one-component case of the full-covariance mixture
 * kernels; the maps M -> scale and d/d scale -> d/d M run per chain in double precision.
 * Workspace: bvi_workspace_bytes(BVI_FAMILY_MVN_TRIL, n, C, d, 0).
 * Replaces the likelihood node MultivariateNormalTriL(loc, scale_tril=cholesky_solve(precision, eye)) of
 * exploratory_experiments/multivariate-gaussian-test.ipynb cell 15 and its gradient. */
int bvi_mvn_tril_logp_grad(const float* Y, int64_t n, int d, int parameterisation, const float* theta, int n_chains,
                           float* logp, float* grad, const double* logp_ref, void* ws, size_t ws_bytes,
                           void* stream);

/* ---- device-side posterior assembly: what tfp.mcmc.TransformedTransitionKernel adds around the
 * target closure (bayes_vi/inference/mcmc/mcmc.py:153-156).  `blocks` is a HOST array.  With
 * theta = f(z) (block-wise constrain), P = sum of sizes, Pu = unconstrained size:
 *   lp[c]     = loglik[c] + sum log prior(theta_c) + forward_log_det_jacobian(z_c)
 *   grad_z[c] = J(z_c)^T (grad_theta[c] + d log prior / d theta) + d fldj / d z
 * bvi_constrain writes theta[C,P]; bvi_posterior_assemble also recomputes theta, ADDS the prior
 * score into grad_theta[C,P] in place, and writes lp[C], grad_z[C,Pu] (grad_theta and grad_z may
 * both be NULL: value only). */
int bvi_constrain(const bvi_param_block* blocks, int n_blocks, int n_chains, const float* z, float* theta,
                  void* stream);
int bvi_posterior_assemble(const bvi_param_block* blocks, int n_blocks, int n_chains, const float* z,
                           float* theta, const float* loglik, float* grad_theta, float* lp, float* grad_z,
                           void* stream);

/* ---- leapfrog / HMC on resident device state: the integrator inside tfp.mcmc.HamiltonianMonteCarlo
 * and NoUTurnSampler (bayes_vi/inference/mcmc/transition_kernels.py:122-128, 238-246) with the
 * log-posterior as (minus) the potential, split at the point where the dataset pass -- and, on
 * several GPUs, the all-reduce of its partial sums -- happens:
 *   pre :  p += eps/2 grad_z;  z += eps p;  theta = f(z)
 *   (caller: loglik, grad_theta = fused likelihood pass at theta [+ all-reduce])
 *   post:  lp, grad_z = assemble(z, loglik, grad_theta);  p += eps/2 grad_z
 * bvi_hmc_begin saves (z, grad_z, lp), draws the momenta p ~ N(0, I) with Philox4x32-10 keyed by
 * (seed; step, chain, element) unless momenta_in is given, and records the kinetic energy;
 * bvi_hmc_end accepts with probability min(1, exp(-dH)) (uniform from the same generator unless
 * log_u_in is given) and restores the saved state of rejected chains; non-finite energies reject.
 * step_offset (device pointer, may be NULL) is added to `step` when the kernel runs: a transition recorded
 * once as a HIP graph draws fresh numbers at every replay if the caller advances that word in between. */
int bvi_leapfrog_pre(const bvi_param_block* blocks, int n_blocks, int n_chains, float step_size, float* z,
                     float* p, const float* grad_z, float* theta, void* stream);
int bvi_leapfrog_post(const bvi_param_block* blocks, int n_blocks, int n_chains, float step_size,
                      const float* z, float* p, float* theta, const float* loglik, float* grad_theta, float* lp,
                      float* grad_z, void* stream);
/* post of one leapfrog step and pre of the next in one launch (the interior steps of a trajectory):
 * same arguments as bvi_leapfrog_post, z and theta are updated as by bvi_leapfrog_pre. */
int bvi_leapfrog_mid(const bvi_param_block* blocks, int n_blocks, int n_chains, float step_size, float* z, float* p,
                     float* theta, const float* loglik, float* grad_theta, float* lp, float* grad_z, void* stream);
int bvi_hmc_begin(int n_chains, int n_unconstrained, uint64_t seed, uint64_t step, const float* z,
                  const float* grad_z, const float* lp, const float* momenta_in, float* p, float* z0, float* g0,
                  float* lp0, float* k0, const uint64_t* step_offset, void* stream);
int bvi_hmc_end(int n_chains, int n_unconstrained, uint64_t seed, uint64_t step, const float* log_u_in,
                float* z, float* grad_z, float* lp, const float* p, const float* z0, const float* g0,
                const float* lp0, const float* k0, unsigned char* is_accepted, float* log_accept_ratio,
                const uint64_t* step_offset, void* stream);

/* Keeps the stored log-densities of all chains small: m = the largest finite lp[c]; lp[c] -= m;
 * *logp_ref += m / n_shards (each of the n_shards row shards subtracts its own *logp_ref inside the likelihood
 * entry points, so the total level moves by m).  lp_absolute (may be NULL) receives float(lp[c] + n_shards *
 * *logp_ref) as it was before the call: the value the reference would report as target_log_prob.  Called between
 * transitions (after bvi_hmc_end / bvi_nuts_subtree_end), when lp holds the chains' current states only.  No
 * counterpart in the reference (TFP keeps float32 log-probs); allocates nothing, capturable in a HIP graph. */
int bvi_logp_rebase(int n_chains, float* lp, int n_shards, double* logp_ref, float* lp_absolute, void* stream);

/* The same for row shards of any sizes.  levels[0] is this shard's own level (the logp_ref of its likelihood calls),
 * levels[1] the level of the total over all shards (identical on every rank); share = rows of this shard / rows of
 * all shards.  m as above; lp[c] -= m; levels[0] += m * share; levels[1] += m; lp_absolute[c] = float(lp[c] +
 * levels[1]) as before the call.  A shard's log-likelihood is its share of the total up to O(sqrt(rows))
 * fluctuations, so each shard's outputs stay small whatever the split (with m / n_shards on unequal shards a
 * 60 / 40 split of 1e7 would leave 1e6 in each shard's floats).  No counterpart in the reference. */
int bvi_logp_rebase_shard(int n_chains, float* lp, double share, double* levels, float* lp_absolute, void* stream);

/* The two leapfrog halves with one signed step size per chain (step_sizes[C]; 0 leaves a chain untouched):
 * what the No-U-Turn sampler needs, whose chains integrate forwards or backwards in time and stop at
 * different tree sizes (transition_kernels.py:238-246). */
int bvi_leapfrog_pre_chainwise(const bvi_param_block* blocks, int n_blocks, int n_chains, const float* step_sizes,
                               float* z, float* p, const float* grad_z, float* theta, void* stream);
int bvi_leapfrog_post_chainwise(const bvi_param_block* blocks, int n_blocks, int n_chains, const float* step_sizes,
                                const float* z, float* p, float* theta, const float* loglik, float* grad_theta,
                                float* lp, float* grad_z, void* stream);

/* ---- the No-U-Turn sampler over the batched likelihood kernels: per-chain tree bookkeeping on the device.
 * Replaces tfp.mcmc.NoUTurnSampler.one_step (bayes_vi/inference/mcmc/transition_kernels.py:238-246) for
 * datasets too large for bvi_nuts_fused.  All arrays are device memory owned by the caller; [C] = n_chains,
 * [C,Pu] = n_chains x n_unconstrained.  Sequence of one transition number `step`:
 *   bvi_nuts_begin;
 *   for depth = 0 .. max_tree_depth-1:  bvi_nuts_subtree_begin(depth);
 *     for leaf = 0 .. 2^depth-1:  bvi_leapfrog_pre_chainwise(se, q, p, gq -> theta), <family>_logp_grad(theta)
 *         [all-reduce over row shards], bvi_leapfrog_post_chainwise(se, q, p, theta, loglik, grad_theta -> lpq, gq),
 *         bvi_nuts_leaf(leaf, any_building)   -- stop the leaf loop when *any_building stays 0
 *     bvi_nuts_subtree_end(any_active)        -- stop the depth loop when *any_active stays 0
 * (the caller zeroes the word before each call that may set it).  Afterwards z, lp, grad hold the new state,
 * sum_acc / n_leap the mean acceptance statistic, flags the per-chain outcome.  Philox streams as in bvi_nuts_fused. */
#define BVI_NUTS_ACTIVE 1
#define BVI_NUTS_BUILDING 2
#define BVI_NUTS_TURNED 4
#define BVI_NUTS_DIVERGED 8
#define BVI_NUTS_MOVED 16
#define BVI_NUTS_FORWARD 32
typedef struct bvi_nuts_state {
  int n_chains, n_unconstrained, max_tree_depth;
  float step_size, max_energy_diff;
  float *z, *lp, *grad;                 /* current sample [C,Pu], its log density [C] and gradient [C,Pu]: in/out */
  float *q, *p, *gq, *lpq;              /* cursor of the subtree under construction [C,Pu] x3, [C] */
  float *ql, *pl, *gl, *qr, *pr, *gr;   /* trajectory ends [C,Pu] */
  float *sub_q, *sub_g, *sub_lp;        /* candidate of the subtree [C,Pu] x2, [C] */
  float *ck_q, *ck_p;                   /* U-turn checkpoints [max_tree_depth+1, C, Pu] */
  float *h0, *log_w, *sub_log_w, *sum_acc, *log_u_sub, *se;   /* [C]; se = signed step size of this subtree, 0 = frozen */
  float *energy, *sub_energy;           /* [C]; log density minus kinetic energy at the selected point / the subtree's
                                           candidate: tfp's kernel_results.energy (out) */
  int *n_leap, *leaf_ctr, *flags;       /* [C]; flags = OR of BVI_NUTS_* */
} bvi_nuts_state;
int bvi_nuts_begin(const bvi_nuts_state* state, uint64_t seed, uint64_t step, void* stream);
int bvi_nuts_subtree_begin(const bvi_nuts_state* state, uint64_t seed, uint64_t step, int depth, void* stream);
int bvi_nuts_leaf(const bvi_nuts_state* state, uint64_t seed, uint64_t step, int leaf, int* any_building, void* stream);
int bvi_nuts_subtree_end(const bvi_nuts_state* state, int* any_active, void* stream);

/* ---- one mean-field ADVI optimisation step around the fused log-posterior pass (what a GradientTape over
 * tfp.vi.monte_carlo_variational_loss, tf.clip_by_norm per variable and tf.optimizers.Adam do in
 * bayes_vi/inference/vi/vi.py:55-75, 109-116 for the surrogate of surrogate_posteriors.py:113-123):
 *   bvi_advi_meanfield_sample   z[S,Pu] = loc + softplus(raw_scale) * eps[S,Pu]
 *   (bvi_constrain, <family>_logp_grad, bvi_posterior_assemble: lp[S] = log pi(z_s), grad_z[S,Pu])
 *   bvi_advi_meanfield_grads    loss = -mean_s (lp_s - log q(z_s)) over the samples where it is finite (vi.py:55-58),
 *                               grad_loc[Pu], grad_raw_scale[Pu] = d loss / d (loc, raw_scale)
 *   bvi_clip_adam_step          per variable: grad <- grad * clip_norm / max(|grad|, clip_norm) (in place; 0 = no
 *                               clipping), then m, v, x updated with lr_t = lr sqrt(1 - b2^t) / (1 - b1^t),
 *                               x -= lr_t m / (sqrt(v) + epsilon); *step_count (device double) is t - 1 on entry
 *                               when advance_step != 0 (first variable of a step) and t otherwise.
 * All arrays are device memory; nothing allocates or synchronises. */
int bvi_advi_meanfield_sample(int n_samples, int n_unconstrained, const float* eps, const float* loc,
                              const float* raw_scale, float* z, void* stream);
int bvi_advi_meanfield_grads(int n_samples, int n_unconstrained, const float* lp, const float* grad_z,
                             const float* eps, const float* raw_scale, float* loss, float* grad_loc,
                             float* grad_raw_scale, void* stream);
/* The full-rank surrogate (scale_tril = tril(raw_tril, -1) + diag(softplus(diag raw_tril) + 1e-5),
 * surrogate_posteriors.py:125-134) in the same two roles; grad_raw_tril[Pu,Pu] is zero above the diagonal. */
int bvi_advi_fullrank_sample(int n_samples, int n_unconstrained, const float* eps, const float* loc,
                             const float* raw_tril, float* z, void* stream);
int bvi_advi_fullrank_grads(int n_samples, int n_unconstrained, const float* lp, const float* grad_z,
                            const float* eps, const float* raw_tril, float* loss, float* grad_loc,
                            float* grad_raw_tril, void* stream);
int bvi_clip_adam_step(int n, float* x, float* grad, float* m, float* v, double* step_count, int advance_step,
                       float learning_rate, float beta_1, float beta_2, float epsilon, float clip_norm,
                       void* stream);

/* Whole HMC transitions in one launch, for datasets small enough that one workgroup per chain can
 * re-stream them at every leapfrog step (tens to ~10^5 rows): per transition a Philox momentum draw,
 * n_leapfrog leapfrog steps with the full-data gradient, and the Metropolis step; n_transitions of them
 * back to back with position, momentum and gradient held in registers.  The same transition, with the
 * same random numbers, as bvi_hmc_begin -> n_leapfrog x (bvi_leapfrog_pre, <family>_logp_grad,
 * bvi_leapfrog_post) -> bvi_hmc_end for transition numbers first_step, first_step + 1, ...; replaces
 * tfp.mcmc.HamiltonianMonteCarlo.one_step iterated by tfp.mcmc.sample_chain
 * (bayes_vi/inference/mcmc/transition_kernels.py:122-128, mcmc.py:160-171).
 *   family      BVI_FAMILY_NORMAL_IID (X unused), BVI_FAMILY_LINREG (X[n,d], d <= 16), BVI_FAMILY_GMM1D (k <= 8)
 *   blocks      the model's parameter blocks; their constrained size must equal the family's parameter count
 *   step_sizes  per-chain step sizes [C], or NULL to use step_size for all chains
 *   z, grad_z, lp   [C,Pu], [C,Pu], [C]: state in, state out; with evaluate_initial != 0 lp and grad_z are
 *               computed from z first (n_transitions may then be 0: a plain evaluation)
 *   momenta_in [T,C,Pu], log_u_in [T,C]   optional injected randomness (tests), else Philox
 *   z_out [T,C,Pu], lp_out [T,C], accepted_out [T,C], log_ratio_out [T,C]   optional per-transition outputs
 * Single GPU only: the rows are not sharded. */
int bvi_hmc_fused(int family, const float* X, const float* y, int64_t n, int d, int k,
                  const bvi_param_block* blocks, int n_blocks, int n_chains, float step_size,
                  const float* step_sizes, int n_leapfrog, int n_transitions, uint64_t seed, uint64_t first_step,
                  int evaluate_initial, float* z, float* grad_z, float* lp, const float* momenta_in,
                  const float* log_u_in, float* z_out, float* lp_out, unsigned char* accepted_out,
                  float* log_ratio_out, void* stream);

/* The No-U-Turn sampler in the same one-workgroup-per-chain form: n_transitions whole transitions per
 * launch, each an iterative tree doubling up to max_tree_depth (<= 10) with multinomial sampling inside
 * subtrees, biased progressive sampling between them, checkpointed U-turn checks and a divergence test
 * against max_energy_diff; replaces tfp.mcmc.NoUTurnSampler.one_step iterated by tfp.mcmc.sample_chain
 * (bayes_vi/inference/mcmc/transition_kernels.py:238-246, mcmc.py:160-171; defaults step_size 0.01,
 * max_tree_depth 5, max_energy_diff 1000 as at transition_kernels.py:148-150).  Arguments as for
 * bvi_hmc_fused; randomness is Philox only (momenta: stream 0; direction and subtree-accept uniform:
 * words 0, 1 of block `depth`, stream 2; multinomial uniform of the n-th leapfrog: word 0 of block n,
 * stream 3).  Per-transition outputs, all optional: z_out [T,C,Pu], lp_out [T,C], moved_out [T,C] (the
 * sample left the initial point), log_accept_out [T,C] (log of the mean min(1, exp(-dH)) over the
 * trajectory: what step-size adaptation consumes), leapfrogs_out [T,C], diverged_out [T,C], energy_out [T,C]
 * (log-density minus kinetic energy at the selected point: tfp's kernel_results.energy, which
 * transition_kernels.py:261 traces). */
int bvi_nuts_fused(int family, const float* X, const float* y, int64_t n, int d, int k,
                   const bvi_param_block* blocks, int n_blocks, int n_chains, float step_size,
                   const float* step_sizes, int max_tree_depth, float max_energy_diff, int n_transitions,
                   uint64_t seed, uint64_t first_step, int evaluate_initial, float* z, float* grad_z, float* lp,
                   float* z_out, float* lp_out, unsigned char* moved_out, float* log_accept_out,
                   int* leapfrogs_out, unsigned char* diverged_out, float* energy_out, void* stream);

/* Random-walk Metropolis in the same form: z' = z + proposal_scale * N(0, I), accepted with probability
 * min(1, pi(z') / pi(z)); replaces tfp.mcmc.RandomWalkMetropolis.one_step with its default proposal
 * (bayes_vi/inference/mcmc/transition_kernels.py:319-321) iterated by tfp.mcmc.sample_chain.  The normal
 * draws and the accept uniform are those of bvi_hmc_begin / bvi_hmc_end for the same transition number.
 * grad_z is carried along (in/out) so that the state stays interchangeable with the other samplers. */
int bvi_rwm_fused(int family, const float* X, const float* y, int64_t n, int d, int k,
                  const bvi_param_block* blocks, int n_blocks, int n_chains, float proposal_scale,
                  int n_transitions, uint64_t seed, uint64_t first_step, int evaluate_initial, float* z,
                  float* grad_z, float* lp, float* z_out, float* lp_out, unsigned char* accepted_out,
                  float* log_ratio_out, void* stream);

/* The generator itself, for tests and for callers that want the same streams: raw Philox4x32-10
 * blocks out[n_chains, n_blocks, 4] for counter (step lo, step hi, chain, block | stream_id << 28)
 * and key (seed lo, seed hi); and the standard normals out[n_chains, n_params] that bvi_hmc_begin
 * draws (stream 0, Box-Muller on 24-bit uniforms). */
int bvi_philox_uint32(uint64_t seed, uint64_t step, int stream_id, int n_chains, int n_blocks, uint32_t* out,
                      void* stream);
int bvi_philox_normal(uint64_t seed, uint64_t step, int n_chains, int n_params, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BVI_H */
