"""fp64 numpy closed forms of the bayes_vi log-posterior hot path (TEST ORACLE).

Test infrastructure only -- see ``oracle/__init__.py``.  Parity against the
TFP outputs of the reference is unpinned (TFP cannot run here).  What pins these
closed forms instead (tests/test_oracle_pins.py, tests/test_host_logic.py):
  normal_iid, linreg : the reference's own result tables -- the NUTS posterior mean / sd of
                       univariate_gaussian_experiment/tables/... and of the 8 datasets of
                       linear_regression_experiment/tables/..., reproduced to their printed decimals by
                       quadrature of these posteriors -- plus scipy.stats / torch.distributions and
                       finite differences;
  gmm1d, gmm_fullcov : scipy.stats / torch.distributions and finite differences only -- the reference
                       holds no table or fixture for the mixtures: PARITY UNPINNED for these two.
  mvn_chol_precision, wishart_tril_prior : scipy.stats (multivariate_normal, wishart) and finite differences; the
                       notebook prints one posterior summary of a 1000-step flow fit on data it does not keep:
                       PARITY UNPINNED.  How TFP reads the arguments (WishartTriL's log_prob of a Cholesky factor
                       without a Jacobian term; MultivariateNormalTriL using the lower triangle of a full matrix)
                       is recalled TFP 0.12 behaviour.

What the reference computes on this path
----------------------------------------
``MCMC.__init__`` (bayes_vi/inference/mcmc/mcmc.py:52-62) and ``VI.__init__``
(bayes_vi/inference/vi/vi.py:40-52) build one closure

    target_log_prob(state) = joint.log_prob(**params, y=targets)

where ``joint`` is ``JointDistributionNamedAutoBatched(priors + likelihood)``
(bayes_vi/model/model.py:171-203): a generative likelihood is wrapped in
``tfd.Sample(likelihood(theta), N)`` (model.py:17-21, 197-198), a regression
likelihood is conditioned on the full feature batch (model.py:194-195).  So

    log pi(theta_c) = sum_p log prior_p(theta_c,p) + sum_{i<N} log p(y_i | theta_c, x_i)

for every chain c (leading batch dimension, mcmc.py:70-72).  Everything here is
written in *constrained* parameter space with a leading chain axis.

Flat parameter layouts (``param_names`` order, each part row-major flattened,
mcmc.py:83-99 / model.py:123-153):

    normal_iid : theta[C, 2]     = [loc, scale]                (README.md:114-120)
    linreg     : theta[C, D+2]   = [beta_0, w_1..w_D, scale]   (README.md:747-756)
    gmm1d      : theta[C, 3K]    = [weights_K, loc_K, scale_K] (univariate-gaussian-mixture-test.ipynb cell 15)
    gmm_fullcov: theta[C, K+K*D+K*D*D] = [weights_K, locs_KxD, scale_tril_KxDxD]
                 (multivariate-gaussian-mixture-density-estimation-test.ipynb cell 7)
    mvn_chol_precision: theta[C, D+D*D] = [loc_D, precision-Cholesky_DxD]
                 (multivariate-gaussian-test.ipynb cell 15)
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import gammaln, logsumexp

LN2PI = math.log(2.0 * math.pi)


def _f64(a):
    return np.asarray(a, dtype=np.float64)


# --------------------------------------------------------------------------
# Priors (README.md:114-117, 747-751; mixture notebook cell 15).
# Each returns (log_prob summed over the event, d log_prob / d x).
# --------------------------------------------------------------------------
def normal_prior(x, loc, scale):
    x, loc, scale = _f64(x), _f64(loc), _f64(scale)
    z = (x - loc) / scale
    lp = -0.5 * z * z - 0.5 * LN2PI - np.log(scale)
    return lp, -z / scale


def halfnormal_prior(x, scale):
    x, scale = _f64(x), _f64(scale)
    lp = 0.5 * math.log(2.0 / math.pi) - np.log(scale) - 0.5 * (x / scale) ** 2
    lp = np.where(x < 0, -np.inf, lp)
    return lp, -x / scale ** 2


def dirichlet_prior(w, concentration):
    """w[..., K] on the simplex; returns (lp[...], dlp/dw[..., K])."""
    w, a = _f64(w), _f64(concentration)
    norm = gammaln(a.sum(-1)) - gammaln(a).sum(-1)
    lp = norm + ((a - 1.0) * np.log(w)).sum(-1)
    return lp, (a - 1.0) / w


def wishart_tril_prior(x, df, scale_tril):
    """tfd.WishartTriL(df, scale_tril, input_output_cholesky=True).log_prob of Cholesky factors
    (multivariate-gaussian-test.ipynb cell 15).  x[..., D*D]: lower-triangular L, row-major; the value is the
    Wishart log density of X = L L^T with V = S S^T,
        (df - D - 1) sum_i log L_ii - |S^-1 L|_F^2 / 2 - [df D log(2) / 2 + df sum_i log S_ii + log Gamma_D(df / 2)],
    with no Jacobian term for L -> X (recalled TFP behaviour).  Returns (lp[...], d lp / d x[..., D*D]) with the
    derivative taken entry by entry of the lower triangle (zero above it)."""
    x, S = _f64(x), _f64(scale_tril)
    d = S.shape[-1]
    L = x.reshape(x.shape[:-1] + (d, d))
    Vinv = np.linalg.inv(S @ S.T)
    diag = np.diagonal(L, axis1=-2, axis2=-1)
    tr = np.einsum('...ij,ik,...kj->...', L, Vinv, L)
    lognorm = (0.5 * df * d * math.log(2.0) + df * np.log(np.diag(S)).sum() + 0.25 * d * (d - 1) * math.log(math.pi)
               + sum(gammaln(0.5 * (df - m)) for m in range(d)))
    lp = (df - d - 1.0) * np.log(diag).sum(-1) - 0.5 * tr - lognorm
    g = -np.einsum('ik,...kj->...ij', Vinv, L)
    idx = np.arange(d)
    g[..., idx, idx] += (df - d - 1.0) / diag
    g = np.tril(g)
    return lp, g.reshape(x.shape)


# --------------------------------------------------------------------------
# Likelihood families: value and gradient of sum_i log p(y_i | theta_c, x_i)
# --------------------------------------------------------------------------
def normal_iid(y, theta):
    """iid Normal likelihood (README.md:119-120 via tfd.Sample, model.py:17-21).

    y[N], theta[C,2] = (loc, scale) -> logp[C], grad[C,2].
    """
    y, theta = _f64(y), _f64(theta)
    loc, sc = theta[:, 0:1], theta[:, 1:2]
    n = y.shape[0]
    r = y[None, :] - loc                      # [C, N]
    s1 = r.sum(-1)
    s2 = (r * r).sum(-1)
    sc1 = sc[:, 0]
    logp = -0.5 * s2 / sc1 ** 2 - n * np.log(sc1) - 0.5 * n * LN2PI
    g_loc = s1 / sc1 ** 2
    g_sc = -n / sc1 + s2 / sc1 ** 3
    return logp, np.stack([g_loc, g_sc], -1)


def linreg(X, y, theta):
    """Gaussian linear regression (README.md:754-756 generalised to D features).

    X[N,D], y[N], theta[C,D+2] = (beta_0, w[D], scale) -> logp[C], grad[C,D+2].
    """
    X, y, theta = _f64(X), _f64(y), _f64(theta)
    n, d = X.shape
    b0, w, sc = theta[:, 0], theta[:, 1:1 + d], theta[:, 1 + d]
    r = y[None, :] - b0[:, None] - w @ X.T     # [C, N]
    s1 = r.sum(-1)
    s2 = (r * r).sum(-1)
    g = r @ X                                  # [C, D]
    logp = -0.5 * s2 / sc ** 2 - n * np.log(sc) - 0.5 * n * LN2PI
    grad = np.concatenate(
        [(s1 / sc ** 2)[:, None], g / sc[:, None] ** 2, (-n / sc + s2 / sc ** 3)[:, None]], -1)
    return logp, grad


def gmm1d(y, theta, k):
    """Univariate Gaussian mixture, MixtureSameFamily(Categorical(probs=w), Normal(loc, scale)).

    y[N], theta[C,3K] = (weights, loc, scale) -> logp[C], grad[C,3K].
    Weight convention: TFP's Categorical(probs=w) takes logits = log(w) and
    normalises them with log_softmax, i.e. uses w_k / sum_j w_j (recalled TFP
    0.12.1 behaviour); the gradient below is of that normalised form.
    """
    y, theta = _f64(y), _f64(theta)
    w, mu, sc = theta[:, :k], theta[:, k:2 * k], theta[:, 2 * k:3 * k]
    n = y.shape[0]
    u = (y[None, None, :] - mu[:, :, None]) / sc[:, :, None]              # [C,K,N]
    logw = np.log(w) - np.log(w.sum(-1, keepdims=True))
    a = logw[:, :, None] - np.log(sc)[:, :, None] - 0.5 * LN2PI - 0.5 * u * u
    lse = logsumexp(a, axis=1)                                            # [C,N]
    rho = np.exp(a - lse[:, None, :])
    logp = lse.sum(-1)
    a0 = rho.sum(-1)
    g_w = a0 / w - n / w.sum(-1, keepdims=True)
    g_mu = (rho * u).sum(-1) / sc
    g_sc = (rho * (u * u - 1.0)).sum(-1) / sc
    return logp, np.concatenate([g_w, g_mu, g_sc], -1)


def gmm_fullcov(Y, weights, locs, tril):
    """Full-covariance mixture, MixtureSameFamily(Categorical(probs), MultivariateNormalTriL).

    Y[N,D], weights[C,K], locs[C,K,D], tril[C,K,D,D] (lower triangular) ->
    logp[C], (g_w[C,K], g_loc[C,K,D], g_tril[C,K,D,D]) with g_tril lower triangular.
    """
    Y, w, mu, L = _f64(Y), _f64(weights), _f64(locs), _f64(tril)
    n, d = Y.shape
    c, k = w.shape
    logw = np.log(w) - np.log(w.sum(-1, keepdims=True))
    logp = np.zeros(c)
    g_w = np.zeros((c, k)); g_mu = np.zeros((c, k, d)); g_L = np.zeros((c, k, d, d))
    for ci in range(c):
        a = np.empty((k, n)); zs = []; Linv = []
        for ki in range(k):
            li = np.linalg.inv(L[ci, ki])
            z = (Y - mu[ci, ki]) @ li.T                                   # [N,D]
            zs.append(z); Linv.append(li)
            a[ki] = (logw[ci, ki] - np.log(np.diag(L[ci, ki])).sum() - 0.5 * d * LN2PI
                     - 0.5 * (z * z).sum(-1))
        lse = logsumexp(a, axis=0)
        rho = np.exp(a - lse[None, :])
        logp[ci] = lse.sum()
        g_w[ci] = rho.sum(-1) / w[ci] - n / w[ci].sum()
        for ki in range(k):
            v = zs[ki] @ Linv[ki]                                         # rows: L^-T z_i
            g_mu[ci, ki] = (rho[ki][:, None] * v).sum(0)
            m = (rho[ki][:, None] * v).T @ zs[ki]                         # sum rho (L^-T z)(z^T)
            g = m - rho[ki].sum() * np.diag(1.0 / np.diag(L[ci, ki]))
            g_L[ci, ki] = np.tril(g)
    return logp, (g_w, g_mu, g_L)


def mvn_chol_precision(Y, theta):
    """MultivariateNormalTriL(loc, scale_tril=cholesky_solve(P, eye)) (multivariate-gaussian-test.ipynb cell 15).

    Y[N,D], theta[C, D+D*D] = (loc, P row-major, lower triangular) -> logp[C], grad[C, D+D*D].
    cholesky_solve(P, I) is the full matrix Sigma = (P P^T)^-1 and MultivariateNormalTriL reads the lower triangle of
    what it is given (recalled TFP behaviour), so the scale is S = tril(Sigma).  Gradient: with G = d logp / d S,
    d logp / d P = -tril(Sigma (G + G^T) Sigma P)."""
    Y, theta = _f64(Y), _f64(theta)
    n, d = Y.shape
    c = theta.shape[0]
    loc, P = theta[:, :d], np.tril(theta[:, d:].reshape(c, d, d))
    Sigma = np.linalg.inv(P @ np.swapaxes(P, -1, -2))
    S = np.tril(Sigma)
    logp, (_, g_mu, g_S) = gmm_fullcov(Y, np.ones((c, 1)), loc[:, None, :], S[:, None])
    G = g_S[:, 0]
    gP = -np.tril(Sigma @ (G + np.swapaxes(G, -1, -2)) @ Sigma @ P)
    return logp, np.concatenate([g_mu[:, 0], gP.reshape(c, d * d)], -1)


def mvn_scale_tril(Y, theta):
    """MultivariateNormalTriL(loc, scale_tril): Y[N,D], theta[C, D+D*D] = (loc, scale_tril) -> logp, grad."""
    Y, theta = _f64(Y), _f64(theta)
    n, d = Y.shape
    c = theta.shape[0]
    loc, S = theta[:, :d], np.tril(theta[:, d:].reshape(c, d, d))
    logp, (_, g_mu, g_S) = gmm_fullcov(Y, np.ones((c, 1)), loc[:, None, :], S[:, None])
    return logp, np.concatenate([g_mu[:, 0], g_S[:, 0].reshape(c, d * d)], -1)


# --------------------------------------------------------------------------
# Constraining bijectors (forward = constrain; model.py:51-55, 112-116).
# TFP's default event-space bijectors are recalled, not observed: Softplus for
# HalfNormal, SoftmaxCentered for Dirichlet, Identity for Normal.
# Each class: forward(x), fldj(x) (summed over the event), vjp(x, g_y) = J^T g_y,
# grad_fldj(x).  The last axis is the event axis.
# --------------------------------------------------------------------------
class Identity:
    def forward(self, x): return _f64(x)
    def inverse(self, y): return _f64(y)
    def fldj(self, x): return np.zeros(_f64(x).shape[:-1])
    def vjp(self, x, g): return _f64(g)
    def grad_fldj(self, x): return np.zeros_like(_f64(x))
    def out_dim(self, n): return n


class Softplus(Identity):
    def forward(self, x): return np.logaddexp(0.0, _f64(x))
    def inverse(self, y):
        y = _f64(y)
        return y + np.log(-np.expm1(-y))
    def fldj(self, x): return (-np.logaddexp(0.0, -_f64(x))).sum(-1)
    def vjp(self, x, g): return _f64(g) / (1.0 + np.exp(-_f64(x)))
    def grad_fldj(self, x): return 1.0 / (1.0 + np.exp(_f64(x)))


class Exp(Identity):
    def forward(self, x): return np.exp(_f64(x))
    def inverse(self, y): return np.log(_f64(y))
    def fldj(self, x): return _f64(x).sum(-1)
    def vjp(self, x, g): return _f64(g) * np.exp(_f64(x))
    def grad_fldj(self, x): return np.ones_like(_f64(x))


class SoftmaxCentered(Identity):
    """R^{K-1} -> simplex: y = softmax([x, 0]); fldj = 0.5 log K + sum_k log y_k."""
    def forward(self, x):
        x = _f64(x)
        xp = np.concatenate([x, np.zeros(x.shape[:-1] + (1,))], -1)
        return np.exp(xp - logsumexp(xp, axis=-1, keepdims=True))
    def inverse(self, y):
        ly = np.log(_f64(y))
        return ly[..., :-1] - ly[..., -1:]
    def fldj(self, x):
        y = self.forward(x)
        return 0.5 * math.log(y.shape[-1]) + np.log(y).sum(-1)
    def vjp(self, x, g):
        y = self.forward(x); g = _f64(g)
        inner = (y * g).sum(-1, keepdims=True)
        return (y * (g - inner))[..., :-1]
    def grad_fldj(self, x):
        y = self.forward(x)
        return 1.0 - y.shape[-1] * y[..., :-1]
    def out_dim(self, n): return n + 1


def fill_triangular_index(d):
    """Index map of TFP's ``fill_triangular`` (lower): position in the D x D
    matrix -> index into the length D(D+1)/2 vector (or -1 above the diagonal).
    Recalled rule: reshape(concat([x[D:], reverse(x)]), [D, D]) then keep the
    lower triangle; e.g. [1..6] -> [[4,0,0],[6,5,0],[3,2,1]].
    """
    m = d * (d + 1) // 2
    idx = np.arange(m)
    full = np.concatenate([idx[d:], idx[::-1]]).reshape(d, d)
    return np.where(np.tril(np.ones((d, d), bool)), full, -1)


class FillScaleTriL(Identity):
    """x[D(D+1)/2] -> lower-triangular [D*D flattened] with softplus(+shift) diagonal."""
    def __init__(self, d, diag_shift=1e-5):
        self.d, self.shift = d, diag_shift
        self.index = fill_triangular_index(d)
    def forward(self, x):
        x = _f64(x); d = self.d
        out = np.zeros(x.shape[:-1] + (d, d))
        for i in range(d):
            for j in range(i + 1):
                v = x[..., self.index[i, j]]
                out[..., i, j] = np.logaddexp(0.0, v) + self.shift if i == j else v
        return out.reshape(x.shape[:-1] + (d * d,))
    def fldj(self, x):
        x = _f64(x)
        diag = np.stack([x[..., self.index[i, i]] for i in range(self.d)], -1)
        return (-np.logaddexp(0.0, -diag)).sum(-1)
    def vjp(self, x, g):
        x = _f64(x); d = self.d
        g = _f64(g).reshape(x.shape[:-1] + (d, d))
        out = np.zeros_like(x)
        for i in range(d):
            for j in range(i + 1):
                v = x[..., self.index[i, j]]
                out[..., self.index[i, j]] = g[..., i, j] / (1.0 + np.exp(-v)) if i == j else g[..., i, j]
        return out
    def grad_fldj(self, x):
        x = _f64(x); out = np.zeros_like(x)
        for i in range(self.d):
            v = x[..., self.index[i, i]]
            out[..., self.index[i, i]] = 1.0 / (1.0 + np.exp(v))
        return out
    def out_dim(self, n): return self.d * self.d


# --------------------------------------------------------------------------
# Whole-model assembly: unconstrained log density and gradient.
# --------------------------------------------------------------------------
class ModelSpec:
    """A restated ``Model`` (model.py:85-169) reduced to what the hot path needs.

    blocks: list of dicts, in ``param_names`` order:
        {'name', 'size' (constrained flat size), 'prior': ('normal', loc, scale) |
         ('halfnormal', scale) | ('dirichlet', conc[K]) | ('wishart_tril', df, scale_tril[D,D]),
         'bijector': instance}
    likelihood: callable theta[C,P] -> (logp[C], grad[C,P]) in constrained space.
    """

    def __init__(self, blocks, likelihood):
        self.blocks, self.likelihood = blocks, likelihood
        self.p = sum(b['size'] for b in blocks)
        self.pu = sum(self._usize(b) for b in blocks)

    @staticmethod
    def _usize(b):
        bij = b['bijector']
        if isinstance(bij, SoftmaxCentered):
            return b['size'] - 1
        if isinstance(bij, FillScaleTriL):
            return bij.d * (bij.d + 1) // 2
        return b['size']

    def constrain(self, z):
        z = _f64(z); out = []; o = 0
        for b in self.blocks:
            n = self._usize(b)
            out.append(b['bijector'].forward(z[:, o:o + n])); o += n
        return np.concatenate(out, -1)

    def unconstrain(self, theta):
        theta = _f64(theta); out = []; o = 0
        for b in self.blocks:
            out.append(b['bijector'].inverse(theta[:, o:o + b['size']])); o += b['size']
        return np.concatenate(out, -1)

    def log_prior(self, theta):
        theta = _f64(theta)
        lp = np.zeros(theta.shape[0]); g = np.zeros_like(theta); o = 0
        for b in self.blocks:
            x = theta[:, o:o + b['size']]; kind = b['prior'][0]
            if kind == 'normal':
                l, d = normal_prior(x, b['prior'][1], b['prior'][2]); l = l.sum(-1)
            elif kind == 'halfnormal':
                l, d = halfnormal_prior(x, b['prior'][1]); l = l.sum(-1)
            elif kind == 'dirichlet':
                l, d = dirichlet_prior(x, b['prior'][1])
            elif kind == 'wishart_tril':
                l, d = wishart_tril_prior(x, b['prior'][1], b['prior'][2])
            else:
                raise ValueError(kind)
            lp += l; g[:, o:o + b['size']] = d; o += b['size']
        return lp, g

    def log_post_constrained(self, theta):
        """log prior + log likelihood and its gradient w.r.t. constrained theta
        (the closure of mcmc.py:57-62)."""
        ll, gl = self.likelihood(theta)
        lp, gp = self.log_prior(theta)
        return ll + lp, gl + gp

    def log_post_unconstrained(self, z):
        """target'(z) = target(f(z)) + fldj(z), as TransformedTransitionKernel
        evaluates it (mcmc.py:153-156), and d/dz."""
        z = _f64(z)
        theta = self.constrain(z)
        lp, g = self.log_post_constrained(theta)
        gz = np.zeros_like(z); oz = 0; ot = 0
        for b in self.blocks:
            n = self._usize(b); bij = b['bijector']
            zz = z[:, oz:oz + n]
            lp = lp + bij.fldj(zz)
            gz[:, oz:oz + n] = bij.vjp(zz, g[:, ot:ot + b['size']]) + bij.grad_fldj(zz)
            oz += n; ot += b['size']
        return lp, gz


def normal_model(y, prior_loc=(0.0, 10.0), prior_scale=10.0):
    """README.md:114-120."""
    blocks = [dict(name='loc', size=1, prior=('normal',) + tuple(prior_loc), bijector=Identity()),
              dict(name='scale', size=1, prior=('halfnormal', prior_scale), bijector=Softplus())]
    return ModelSpec(blocks, lambda th: normal_iid(y, th))


def linreg_model(X, y, prior_sd=10.0, prior_scale=10.0):
    """README.md:747-756 with a D-vector of slopes."""
    d = np.asarray(X).shape[1]
    blocks = [dict(name='beta_0', size=1, prior=('normal', 0.0, prior_sd), bijector=Identity()),
              dict(name='w', size=d, prior=('normal', 0.0, prior_sd), bijector=Identity()),
              dict(name='scale', size=1, prior=('halfnormal', prior_scale), bijector=Softplus())]
    return ModelSpec(blocks, lambda th: linreg(X, y, th))


def gmm1d_model(y, k, prior_sd=10.0, prior_scale=10.0):
    """univariate-gaussian-mixture-test.ipynb cell 15."""
    blocks = [dict(name='weights', size=k, prior=('dirichlet', np.ones(k)), bijector=SoftmaxCentered()),
              dict(name='loc', size=k, prior=('normal', 0.0, prior_sd), bijector=Identity()),
              dict(name='scale', size=k, prior=('halfnormal', prior_scale), bijector=Softplus())]
    return ModelSpec(blocks, lambda th: gmm1d(y, th, k))


def mvn_chol_precision_model(Y, df=5.0, prior_sd=10.0, wishart_scale=10.0, diag_shift=1e-3):
    """multivariate-gaussian-test.ipynb cell 15: loc ~ MultivariateNormalDiag(0, 10), precision ~
    WishartTriL(df=5, scale_tril=10 I, input_output_cholesky=True), bijectors Identity and
    FillScaleTriL(Softplus, diag_shift=1e-3)."""
    d = np.asarray(Y).shape[1]
    blocks = [dict(name='loc', size=d, prior=('normal', 0.0, prior_sd), bijector=Identity()),
              dict(name='precision', size=d * d, prior=('wishart_tril', df, wishart_scale * np.eye(d)),
                   bijector=FillScaleTriL(d, diag_shift))]
    return ModelSpec(blocks, lambda th: mvn_chol_precision(Y, th))
