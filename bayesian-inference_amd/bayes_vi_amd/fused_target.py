"""The drop-in for the reference's `target_log_prob` closure.

In the reference, `MCMC.__init__` (bayes_vi/inference/mcmc/mcmc.py:52-62) and
`VI.__init__` (bayes_vi/inference/vi/vi.py:40-52) capture the whole dataset as
one batch and build `target_log_prob(state) = joint.log_prob(**state, y=targets)`;
TFP then evaluates that closure and its autodiff gradient for all chains at
every leapfrog step / ELBO sample.  Here the same quantity is split into

  * `FusedLogLikelihood`: sum over the (local shard of) datapoints of the
    log-likelihood and its gradient for all chains, one fused HIP pass through
    the C ABI (`_native`), all-reduced over the row shards of a process group;
  * `Posterior`: + log-priors and, in unconstrained space, the bijector chain
    rule and forward-log-det-Jacobian terms -- O(chains x parameters) torch ops
    added once, after the reduction.

There is no CPU path: tensors must live on a GPU and libbvi_hip.so must load.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _native
from . import bijectors as _bij
from . import distributions as _dist
from .bijectors import Blockwise


class ShardedLogLikelihood:
    """Value and gradient of sum_i log p(y_i | theta_c, x_i) for all chains c, where this
    process holds one row shard of the dataset.

    `group`: a torch.distributed process group whose ranks each hold a disjoint row shard;
    the [C, 1+P] partial sums (log-likelihood and gradient, one flat buffer) are summed over
    it with ONE all-reduce per evaluation (RCCL on GPUs, gloo in CPU tests).  Chains are
    replicated: every rank evaluates the same theta and ends up with the same totals.
    `None` = single shard.  Subclasses implement `_call` (the local partial sums).
    `fixed_order_reduction = True` replaces the all-reduce by an all-gather of the partials and a
    local sum in rank order: the totals then do not depend on the collective's algorithm (bitwise
    reproducible across library versions and topologies) at the price of world_size x the bytes.
    """
    family = None
    n_params = None
    fixed_order_reduction = False

    def __init__(self, device, group=None):
        self.device = torch.device(device)
        self.group = group
        self.sharded = group is not None and dist.is_initialized() and dist.get_world_size(group) > 1

    def _call(self, theta, logp, grad):
        raise NotImplementedError

    def logp_grad(self, theta, want_grad=True, relative=False):
        """theta[C, P] float32 (constrained) -> (logp[C], grad[C, P] or None), summed over ALL shards.
        `relative=True` (the device samplers only) asks for log-likelihoods relative to the level the likelihood
        keeps on the device (`FusedLogLikelihood.logp_ref`); every other caller gets absolute values."""
        theta = theta.contiguous()
        c = theta.shape[0]
        # one flat buffer so that the shard reduction is a single collective
        buf = torch.empty(c * (1 + (self.n_params if want_grad else 0)), dtype=torch.float32, device=theta.device)
        logp = buf[:c]
        grad = buf[c:].view(c, self.n_params) if want_grad else None
        if relative:
            self._call(theta, logp, grad, relative=True)
        else:
            self._call(theta, logp, grad)
        if self.sharded:
            # a gloo group cannot reduce device memory (it waits for ever): such a group -- a rehearsal of the
            # multi-rank path on fewer GPUs than ranks -- gets the buffer staged through the host
            staged = buf.is_cuda and dist.get_backend(self.group) == 'gloo'
            wire = buf.cpu() if staged else buf
            if self.fixed_order_reduction:
                world = dist.get_world_size(self.group)
                parts = torch.empty(world * wire.numel(), dtype=wire.dtype, device=wire.device)
                dist.all_gather_into_tensor(parts, wire, group=self.group)
                total = parts[:wire.numel()].clone()
                for r in range(1, world):           # rank order, the same on every rank
                    total += parts[r * wire.numel():(r + 1) * wire.numel()]
                buf.copy_(total)
            else:
                dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group)
                if staged:
                    buf.copy_(wire)
        return logp, grad


def replicate_from_rank0(tensor, group):
    """Every rank of a row-sharded run must hold the same chains / surrogate parameters / random stream; this makes
    `tensor` on all ranks equal to rank 0's, in place (a gloo group gets device memory staged through the host)."""
    if group is None or not dist.is_initialized() or dist.get_world_size(group) <= 1:
        return tensor
    src = dist.get_global_rank(group, 0) if hasattr(dist, 'get_global_rank') else 0
    if tensor.is_cuda and dist.get_backend(group) == 'gloo':
        wire = tensor.detach().cpu()
        dist.broadcast(wire, src=src, group=group)
        with torch.no_grad():
            tensor.copy_(wire)
    else:
        dist.broadcast(tensor.detach(), src=src, group=group)
    return tensor


class FusedLogLikelihood(ShardedLogLikelihood):
    """The local partial sums come from one fused HIP pass through the C ABI.  MI355X only."""

    def __init__(self, device, group=None):
        super().__init__(device, group)
        if self.device.type != 'cuda':
            raise _native.BviError('the fused log-likelihood runs on MI355X only; got device %s' % device)
        _native.lib()
        self.ws = _native.Workspace(self.device)
        # level (one float64 on the device) that the kernels subtract from a log-likelihood before rounding it to
        # float, for the calls that ask for it (`logp_grad(..., relative=True)`: the device samplers, which keep it
        # near the chains' current values with Posterior.rebase so that the differences they decide on stay exact
        # when |log-likelihood| ~ 1e7).  Calls without `relative` return absolute values whether a level exists or not.
        self.logp_ref = None

    def _ref(self, relative):
        return self.logp_ref.data_ptr() if relative and self.logp_ref is not None else None


def _launch(fn, *args):
    _native.check(fn(*args), fn.__name__)


class NormalIID(FusedLogLikelihood):
    """y_i ~ Normal(loc, scale) iid; theta = (loc, scale).  README.md:114-120."""
    family = _native.FAMILY_NORMAL_IID
    n_params = 2

    def __init__(self, y, group=None):
        super().__init__(y.device, group)
        self.y = y.to(torch.float32).contiguous().reshape(-1)
        self.n = self.y.shape[0]

    def _call(self, theta, logp, grad, relative=False):
        c = theta.shape[0]
        buf = self.ws.get(_native.workspace_bytes(self.family, self.n, c))
        _launch(_native.lib().bvi_normal_iid_logp_grad, _native._ptr(self.y), self.n, _native._ptr(theta), c,
                _native._ptr(logp), _native._ptr(grad), self._ref(relative), _native._ptr(buf), buf.numel(), _native._stream())


class LinearRegression(FusedLogLikelihood):
    """y_i ~ Normal(beta_0 + x_i . w, scale); theta = (beta_0, w[D], scale).  README.md:747-756."""
    family = _native.FAMILY_LINREG

    def __init__(self, X, y, group=None, precision=_native.PREC_AUTO, pack=True):
        super().__init__(X.device, group)
        self.X = X.to(torch.float32).contiguous()
        if self.X.dim() == 1:
            self.X = self.X[:, None]
        self.y = y.to(torch.float32).contiguous().reshape(-1)
        # the kernels read 16-byte vectors: a row slice of a larger tensor (a shard) may start between them
        if self.X.data_ptr() % 16:
            self.X = self.X.clone()
        if self.y.data_ptr() % 16:
            self.y = self.y.clone()
        self.n, self.d = self.X.shape
        self.n_params = self.d + 2
        self.precision = precision
        self.pack = pack
        self.packed = None        # piece planes for a matrix-core kernel, built once on first use
        self.packed_format = None

    def matrix_core_format(self, c):
        """Packed format the evaluation of `c` chains runs on, or None for the exact-f32 kernels.
        AUTO: the f16-plane kernel (log-density at f32 accuracy, gradient error at the level of the f32 kernel's)
        for more than 16 chains (32 for at most 8 features, where the vector-ALU kernels stream less data) on a
        shard of at least 16384 rows, and for any number of chains from 400 000 rows of more than 8 features on:
        the plane is half the bytes of the f32 matrix (N = 10M, D = 64: 0.26 ms against 0.56 ms of the exact-f32
        kernel for 1 - 16 chains; 500k rows 0.040 against 0.046 ms; at 300k rows and below the two are within
        launch latency of each other).
        BF16X3 / BF16X6 select the bf16-piece kernel explicitly."""
        if not self.pack:
            return None
        if self.precision in (_native.PREC_BF16X3, _native.PREC_BF16X6):
            return _native.PACK_BF16X3 if self.d in (32, 64) else None
        if self.precision != _native.PREC_AUTO or self.d > 64:
            return None
        return _native.linreg_recommended_format(self.n, self.d, c)      # the library's own rule (include/bvi.h)

    def _call(self, theta, logp, grad, relative=False):
        c = theta.shape[0]
        buf = self.ws.get(_native.workspace_bytes(self.family, self.n, c, d=self.d))
        fmt = self.matrix_core_format(c)
        if fmt is not None:
            # the dataset is immutable after construction (mcmc.py:52): split it into the 16-bit
            # pieces of the matrix-core kernel once, then every evaluation stages tiles by plain copies
            if self.packed is None or self.packed_format != fmt:
                self.packed = _native.linreg_pack(self.X, self.y, fmt=fmt)
                self.packed_format = fmt
            _launch(_native.lib().bvi_linreg_packed_logp_grad, _native._ptr(self.packed), self.n, self.d,
                    fmt, _native._ptr(theta), c, self.precision, _native._ptr(logp), _native._ptr(grad), self._ref(relative),
                    _native._ptr(buf), buf.numel(), _native._stream())
            return
        _launch(_native.lib().bvi_linreg_logp_grad, _native._ptr(self.X), _native._ptr(self.y), self.n, self.d,
                _native._ptr(theta), c, self.precision, _native._ptr(logp), _native._ptr(grad), self._ref(relative),
                _native._ptr(buf), buf.numel(), _native._stream())


class GaussianMixture1D(FusedLogLikelihood):
    """y_i ~ sum_k w_k Normal(loc_k, scale_k); theta = (weights[K], loc[K], scale[K]).
    exploratory_experiments/univariate-gaussian-mixture-test.ipynb cell 15."""
    family = _native.FAMILY_GMM1D

    def __init__(self, y, k, group=None, sort_rows=True):
        super().__init__(y.device, group)
        self.y = y.to(torch.float32).contiguous().reshape(-1)
        if sort_rows:
            # The rows are exchangeable, so their order is the layout's to choose: in ascending order neighbouring rows
            # lie within reach of the same few components, which the kernels whose lanes hold several rows (up to 16
            # chains) use to leave the others out (N = 100M, K = 8: 4 chains, value only 0.61 -> 0.34 ms; 16 chains
            # 3.9 -> 2.9 ms).  Sums change in the last float digits only.
            self.y = torch.sort(self.y).values.contiguous()
        self.n, self.k = self.y.shape[0], int(k)
        self.n_params = 3 * self.k

    def _call(self, theta, logp, grad, relative=False):
        c = theta.shape[0]
        buf = self.ws.get(_native.workspace_bytes(self.family, self.n, c, k=self.k))
        _launch(_native.lib().bvi_gmm1d_logp_grad, _native._ptr(self.y), self.n, self.k, _native._ptr(theta), c,
                _native._ptr(logp), _native._ptr(grad), self._ref(relative), _native._ptr(buf), buf.numel(), _native._stream())


def cluster_centres(k):
    """Centres the row ordering of a k-component mixture works with: 4 k, at least 16."""
    return int(min(_native.CLUSTER_MAX_CENTRES, max(16, 4 * int(k))))


def cluster_row_order(Y, k):
    """A permutation of the rows of Y[N, D] that puts rows lying close together next to each other
    (`bvi_cluster_rows`: Lloyd's algorithm on 4 k centres started from rows at a fixed stride, rows ordered by the
    centre they end up with, reproducible bit for bit).  The rows of a likelihood sum are exchangeable, so their order
    in memory is the layout's to choose; the full-covariance mixture kernels leave a component out of every 64-row
    tile it cannot reach, which is most tiles once neighbouring rows belong to the same few components.  Cost: a
    handful of [N, 4 k] distance passes, once per dataset."""
    return _native.cluster_rows(Y.to(torch.float32).contiguous(), cluster_centres(k), want_order=True)[1]


class GaussianMixtureFullCov(FusedLogLikelihood):
    """y_i ~ sum_k w_k N(mu_k, L_k L_k^T); theta = (weights[K], locs[K, D], scale_tril[K, D, D]).
    exploratory_experiments/multivariate-gaussian-mixture-density-estimation-test.ipynb cell 7.
    `cluster_rows` (default): this object's copy of the rows is stored cluster by cluster (`cluster_row_order`);
    the sums do not depend on the order beyond float rounding, the kernel's speed does."""
    family = _native.FAMILY_GMM_FULLCOV

    def __init__(self, Y, k, group=None, cluster_rows=True):
        super().__init__(Y.device, group)
        self.Y = Y.to(torch.float32).contiguous()
        if cluster_rows and self.Y.shape[0] >= 4096:
            self.Y = _native.cluster_rows(self.Y, cluster_centres(k))
        self.n, self.d = self.Y.shape
        self.k = int(k)
        self.n_params = self.k * (1 + self.d + self.d * self.d)
        # what the kernels need of the rows alone (statistics, tile spheres, f16 pieces): once, not per evaluation
        self.packed = _native.gmm_fullcov_pack(self.Y) if self.n > 0 else None

    def _call(self, theta, logp, grad, relative=False):
        c = theta.shape[0]
        buf = self.ws.get(_native.workspace_bytes(self.family, self.n, c, d=self.d, k=self.k))
        if self.packed is None:
            _launch(_native.lib().bvi_gmm_fullcov_logp_grad, _native._ptr(self.Y), self.n, self.d, self.k,
                    _native._ptr(theta), c, _native._ptr(logp), _native._ptr(grad), self._ref(relative),
                    _native._ptr(buf), buf.numel(), _native._stream())
            return
        _launch(_native.lib().bvi_gmm_fullcov_packed_logp_grad, _native._ptr(self.Y), _native._ptr(self.packed), self.n,
                self.d, self.k, _native._ptr(theta), c, _native._ptr(logp), _native._ptr(grad), self._ref(relative),
                _native._ptr(buf), buf.numel(), _native._stream())


class MultivariateNormalTriL(FusedLogLikelihood):
    """y_i ~ N(loc, S S^T); theta = (loc[D], M[D, D]) with S = tril(M) (`MVN_SCALE_TRIL`) or, for
    `MVN_CHOL_PRECISION_SOLVE`, S = tril((M M^T)^-1): what MultivariateNormalTriL(loc, scale_tril=
    cholesky_solve(M, eye)) evaluates.  exploratory_experiments/multivariate-gaussian-test.ipynb cell 15."""
    family = _native.FAMILY_MVN_TRIL

    def __init__(self, Y, parameterisation=_native.MVN_SCALE_TRIL, group=None):
        super().__init__(Y.device, group)
        self.Y = Y.to(torch.float32).contiguous()
        self.n, self.d = self.Y.shape
        self.parameterisation = int(parameterisation)
        self.n_params = self.d + self.d * self.d
        self.packed = _native.gmm_fullcov_pack(self.Y) if self.n > 0 else None

    def _call(self, theta, logp, grad, relative=False):
        c = theta.shape[0]
        buf = self.ws.get(_native.workspace_bytes(self.family, self.n, c, d=self.d))
        if self.packed is None:
            _launch(_native.lib().bvi_mvn_tril_logp_grad, _native._ptr(self.Y), self.n, self.d, self.parameterisation,
                    _native._ptr(theta), c, _native._ptr(logp), _native._ptr(grad), self._ref(relative),
                    _native._ptr(buf), buf.numel(), _native._stream())
            return
        _launch(_native.lib().bvi_mvn_tril_packed_logp_grad, _native._ptr(self.Y), _native._ptr(self.packed), self.n,
                self.d, self.parameterisation, _native._ptr(theta), c, _native._ptr(logp), _native._ptr(grad),
                self._ref(relative), _native._ptr(buf), buf.numel(), _native._stream())


class Posterior:
    """log pi(theta) = log-likelihood (fused) + sum of log-priors, in constrained
    and unconstrained coordinates.

    priors: list of (distribution, flat_size) in parameter order; bijectors: one
    per prior (forward = constrain).  Flat layouts follow the reference
    (param_names order, each part row-major flattened: mcmc.py:83-99).
    """

    def __init__(self, likelihood: ShardedLogLikelihood, priors, bijectors):
        self.likelihood = likelihood
        self.priors = [p for p, _ in priors]
        self.csizes = [int(s) for _, s in priors]
        self.usizes = [b.inverse_event_size(s) for b, s in zip(bijectors, self.csizes)]
        self.bijector = Blockwise(bijectors, self.usizes)
        self.n_constrained = sum(self.csizes)
        self.n_unconstrained = sum(self.usizes)
        if self.n_constrained != likelihood.n_params:
            raise ValueError('priors describe %d parameters, the likelihood takes %d'
                             % (self.n_constrained, likelihood.n_params))

    def device_blocks(self):
        """The model as an array of bvi_param_block for the device-side assembly / leapfrog entry
        points, or None when a prior or bijector is outside what they implement (per-element
        hyper-parameters, user bijectors): the torch path below then does the same job."""
        if getattr(self, '_blocks', False) is not False:
            return self._blocks
        specs = []
        for prior, bij, size in zip(self.priors, self.bijector.bijectors, self.csizes):
            kinds = {_bij.Identity: _native.BIJ_IDENTITY, _bij.Softplus: _native.BIJ_SOFTPLUS,
                     _bij.Exp: _native.BIJ_EXP, _bij.SoftmaxCentered: _native.BIJ_SOFTMAX_CENTERED,
                     _bij.FillScaleTriL: _native.BIJ_FILL_SCALE_TRIL}
            if type(bij) not in kinds:
                specs = None
                break

            def scalar(t):
                t = torch.as_tensor(t, dtype=torch.float32).reshape(-1)
                return float(t[0]) if bool((t == t[0]).all()) else None
            if isinstance(prior, _dist.Normal):
                a, b = scalar(prior.loc), scalar(prior.scale)
                spec = (_native.PRIOR_NORMAL, a, b)
            elif isinstance(prior, _dist.HalfNormal):
                spec = (_native.PRIOR_HALFNORMAL, scalar(prior.scale), 0.0)
            elif isinstance(prior, _dist.Dirichlet):
                spec = (_native.PRIOR_DIRICHLET, scalar(prior.concentration), 0.0)
            elif isinstance(prior, _dist.MultivariateNormalDiag):
                spec = (_native.PRIOR_NORMAL, scalar(prior.loc), scalar(prior.scale_diag))
            elif isinstance(prior, _dist.WishartTriL):
                # the device prior covers scale_tril = s I (what the reference's model uses); others go through torch
                spec = (_native.PRIOR_WISHART_TRIL, float(prior.df), prior.isotropic_scale())
                if not isinstance(bij, _bij.FillScaleTriL) or bij.d != prior.d:
                    spec = (None, None, None)
            else:
                spec = (None, None, None)
            if spec[0] is None or spec[1] is None or spec[2] is None:
                specs = None
                break
            if isinstance(bij, _bij.FillScaleTriL):
                if spec[0] == _native.PRIOR_DIRICHLET or size != bij.batch * bij.d * bij.d:
                    specs = None
                    break
                specs.append((kinds[type(bij)], size, spec[0], spec[1], spec[2], bij.d, float(bij.diag_shift)))
            else:
                specs.append((kinds[type(bij)], size, spec[0], spec[1], spec[2]))
        if specs is not None and len(specs) > _native.MAX_PARAM_BLOCKS:
            specs = None
        self._blocks = _native.make_blocks(specs) if specs is not None else None
        return self._blocks

    # ---- log-densities relative to a level that follows the chains ----
    # A float holds log pi ~ 1e7 (10M rows) to +-0.5, the size of the energy differences the samplers decide on.
    # The device samplers therefore ask the likelihood kernels for values relative to a level kept on the device
    # (one float64; bvi_logp_rebase moves it to the chains' best current value after every transition), so what is
    # stored and subtracted stays small and exact; target_log_prob is reported absolute, as the reference does.
    def use_relative_log_prob(self):
        """Give the likelihood a level: two float64 on the device, [this row shard's own level, the level of the
        total over all shards].  Only calls that ask for it (`relative=True`) subtract it."""
        lik = self.likelihood
        if hasattr(lik, 'logp_ref') and lik.logp_ref is None and lik.device.type == 'cuda':
            lik.logp_ref = torch.zeros(2, dtype=torch.float64, device=lik.device)
        return getattr(lik, 'logp_ref', None) is not None

    def _share(self):
        """This shard's part of the rows (1 without shards): the part of the total level its own level takes.  A
        shard's log-likelihood is that part of the total up to O(sqrt(rows)), whatever the sizes of the shards."""
        if getattr(self, '_row_share', None) is None:
            lik = self.likelihood
            share = 1.0
            if lik.sharded:
                inner = getattr(lik, 'inner', lik)
                rows = getattr(inner, 'n', None)
                world = dist.get_world_size(lik.group)
                if rows is None:
                    share = 1.0 / world
                else:
                    on_host = dist.get_backend(lik.group) == 'gloo'
                    t = torch.tensor([float(rows)], dtype=torch.float64, device='cpu' if on_host else lik.device)
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=lik.group)
                    share = float(rows) / float(t[0])
            self._row_share = share
        return self._row_share

    def rebase(self, lp, lp_absolute=None):
        """After a transition: re-level the chains' stored log-densities `lp` in place (no-op without a level);
        lp_absolute, if given, receives their absolute values."""
        ref = getattr(self.likelihood, 'logp_ref', None)
        if ref is None:
            if lp_absolute is not None:
                lp_absolute.copy_(lp)
            return
        _native.logp_rebase_shard(lp, self._share(), ref, lp_absolute)

    def initial_log_prob_and_grad(self, z, want_grad=True):
        """First evaluation of a run: with a level in use, one pass finds it (the values of that pass are absolute
        floats, +-0.5 at |log pi| ~ 1e7), the level moves to the best chain, and a second pass returns values
        relative to it -- so that a chain which rejects its first proposals does not keep a rounded log-density."""
        lp, g = self.unconstrained_log_prob_and_grad(z, want_grad, relative=True)
        if getattr(self.likelihood, 'logp_ref', None) is not None:
            self.rebase(lp.contiguous())
            lp, g = self.unconstrained_log_prob_and_grad(z, want_grad, relative=True)
        return lp, g

    def state_log_prob(self, state):
        """The relative log-densities a sampler works on, from a sampler state.  States made by `log_prob_state`
        carry them together with the level they refer to; the level may have moved since (another kernel on the
        same posterior), which is corrected in double.  A bare state (absolute values only) is converted."""
        ref = getattr(self.likelihood, 'logp_ref', None)
        if ref is None:
            return state['target_log_prob'].clone()
        level = ref[1]
        if 'relative_log_prob' in state:
            return (state['relative_log_prob'].double() + (state['log_prob_level'] - level)).float()
        return (state['target_log_prob'].double() - level).float()

    def log_prob_state(self, lp, lp_absolute=None):
        """State entries for relative log-densities `lp` under the current level."""
        ref = getattr(self.likelihood, 'logp_ref', None)
        if ref is None:
            return {'target_log_prob': lp}
        return {'target_log_prob': self.absolute(lp) if lp_absolute is None else lp_absolute,
                'relative_log_prob': lp, 'log_prob_level': ref[1].clone()}

    def absolute(self, lp):
        """Absolute log-densities for values expressed relative to the current level."""
        ref = getattr(self.likelihood, 'logp_ref', None)
        if ref is None:
            return lp
        return (lp.double() + ref[1]).float()

    # ---- constrained space: the closure of mcmc.py:57-62 / vi.py:46-52 ----
    def log_prior(self, theta, want_grad=True):
        parts = torch.split(theta, self.csizes, dim=-1)
        lp = sum(p.log_prob_flat(x) for p, x in zip(self.priors, parts))
        g = torch.cat([p.score_flat(x) for p, x in zip(self.priors, parts)], -1) if want_grad else None
        return lp, g

    def log_prob_and_grad(self, theta, want_grad=True):
        ll, gl = self.likelihood.logp_grad(theta, want_grad)
        lp, gp = self.log_prior(theta, want_grad)
        return ll + lp, (gl + gp if want_grad else None)

    def log_prob(self, theta):
        return self.log_prob_and_grad(theta, want_grad=False)[0]

    # ---- unconstrained space: what TransformedTransitionKernel evaluates (mcmc.py:153-156) ----
    def unconstrained_log_prob_and_grad(self, z, want_grad=True, relative=False):
        """log pi(z) = log pi(f(z)) + fldj(z) and its gradient in unconstrained space, z[C, Pu]: what
        tfp.mcmc.TransformedTransitionKernel makes of the target closure (mcmc.py:153-156).  With built-in
        priors and bijectors on the GPU this is three launches -- bvi_constrain, the fused data pass,
        bvi_posterior_assemble --; other models compose the same quantities from the bijector objects.
        Values are absolute unless `relative` (device samplers: relative to the likelihood's level)."""
        blocks = self.device_blocks() if z.is_cuda and z.dtype == torch.float32 and z.dim() == 2 else None
        if blocks is not None:
            lib, ptr, st = _native.lib(), _native._ptr, _native._stream()
            z = z.contiguous()
            c = z.shape[0]
            theta = torch.empty(c, self.n_constrained, dtype=torch.float32, device=z.device)
            _native.check(lib.bvi_constrain(blocks, len(blocks), c, ptr(z), ptr(theta), st), 'bvi_constrain')
            ll, gl = self.likelihood.logp_grad(theta, want_grad, relative) if relative else \
                self.likelihood.logp_grad(theta, want_grad)
            lp = torch.empty(c, dtype=torch.float32, device=z.device)
            gz = torch.empty_like(z) if want_grad else None
            _native.check(lib.bvi_posterior_assemble(blocks, len(blocks), c, ptr(z), ptr(theta), ptr(ll),
                                                     ptr(gl) if want_grad else None, ptr(lp), ptr(gz), st),
                          'bvi_posterior_assemble')
            return lp, gz
        theta = self.bijector.forward(z)
        lp, g = self.log_prob_and_grad(theta, want_grad)
        lp = lp + self.bijector.forward_log_det_jacobian(z)
        if not want_grad:
            return lp, None
        return lp, self.bijector.vjp(z, g) + self.bijector.grad_fldj(z)

    def unconstrained_log_prob(self, z):
        return self.unconstrained_log_prob_and_grad(z, want_grad=False)[0]


class _TargetLogProbFn(torch.autograd.Function):
    """Makes the fused pass one autograd node: forward computes log pi AND keeps
    d log pi / d theta from the same data pass, backward is an O(S x P) product.
    (The reference's GradientTape re-traverses the [S, N] graph: vi.py:66-69.)"""

    @staticmethod
    def forward(ctx, theta, posterior):
        lp, g = posterior.log_prob_and_grad(theta.detach(), want_grad=True)
        # a sample counts only if its log density AND its whole gradient are finite (a scale underflowing towards 0
        # gives log pi = -1e38 with d log pi = -inf): such a sample reports -inf, so that the finite-mask of
        # vi.py:55-58 drops it from loss and gradient alike instead of turning every variable into NaN
        lp = torch.where(torch.isfinite(g).all(-1), lp, torch.full_like(lp, float('-inf')))
        ctx.save_for_backward(g)
        return lp

    @staticmethod
    def backward(ctx, grad_out):
        g, = ctx.saved_tensors
        # a sample whose discrepancy was dropped as non-finite (vi.py:55-58) arrives with weight 0 and may carry a
        # non-finite gradient: it must contribute nothing, not 0 * inf
        w = grad_out[:, None]
        return torch.where(w == 0, torch.zeros_like(g), w * g), None


def differentiable_log_prob(posterior: Posterior, theta):
    """Constrained log pi(theta) for theta[S, P], differentiable by torch autograd."""
    return _TargetLogProbFn.apply(theta, posterior)


class PotentialEnergy:
    """U(q) = -log pi(q) in unconstrained coordinates as an energy function for
    `LeapfrogIntegrator.solve` (bayes_vi/utils/symplectic_integrators.py:12-59): calling it gives the energy
    [C] (differentiable by autograd through the fused node), `.grad(q)` its gradient from the fused pass without
    autograd.  Together with `QuadraticKineticEnergy` the integrator runs its steps on the device-resident
    leapfrog entry points (`leapfrog_on_device`)."""

    def __init__(self, posterior: Posterior):
        self.posterior = posterior

    def __call__(self, q):
        bij = self.posterior.bijector
        return -(differentiable_log_prob(self.posterior, bij.forward(q).contiguous())
                 + bij.forward_log_det_jacobian(q))

    def grad(self, q):
        return -self.posterior.unconstrained_log_prob_and_grad(q.detach().contiguous())[1]


class QuadraticKineticEnergy:
    """T(p) = |p|^2 / 2, the kinetic energy of HMC with unit mass."""

    def __call__(self, p):
        return 0.5 * (p * p).sum(-1)

    def grad(self, p):
        return p


def leapfrog_on_device(posterior: Posterior, state, step_size: float, n_steps: int):
    """n leapfrog steps of H(q, p) = -log pi(q) + |p|^2 / 2 on state[C, 2 Pu] = [q | p] with q, p and the
    gradient resident on the device: bvi_constrain + data pass + bvi_posterior_assemble for the gradient at the
    start, then per step bvi_leapfrog_pre -> fused data pass (+ all-reduce over row shards) -> bvi_leapfrog_mid
    (bvi_leapfrog_post after the last step).  The same update order as symplectic_integrators.py:45-56 with the
    half steps of neighbouring iterations merged.  Not differentiable; returns a new [C, 2 Pu] tensor."""
    blocks = posterior.device_blocks()
    if blocks is None:
        raise _native.BviError('the device leapfrog needs built-in priors and bijectors')
    lib, ptr, st = _native.lib(), _native._ptr, _native._stream()
    pu, pc = posterior.n_unconstrained, posterior.n_constrained
    c = state.shape[0]
    z = state[:, :pu].detach().clone().contiguous()
    p = state[:, pu:].detach().clone().contiguous()
    if n_steps <= 0:
        return torch.cat([z, p], -1)
    nb = len(blocks)
    theta = torch.empty(c, pc, dtype=torch.float32, device=z.device)
    lp = torch.empty(c, dtype=torch.float32, device=z.device)
    g = torch.empty_like(z)
    _native.check(lib.bvi_constrain(blocks, nb, c, ptr(z), ptr(theta), st), 'bvi_constrain')
    ll, gl = posterior.likelihood.logp_grad(theta, True)
    _native.check(lib.bvi_posterior_assemble(blocks, nb, c, ptr(z), ptr(theta), ptr(ll), ptr(gl), ptr(lp), ptr(g), st),
                  'bvi_posterior_assemble')
    eps = float(step_size)
    _native.check(lib.bvi_leapfrog_pre(blocks, nb, c, eps, ptr(z), ptr(p), ptr(g), ptr(theta), st), 'bvi_leapfrog_pre')
    for k in range(n_steps):
        ll, gl = posterior.likelihood.logp_grad(theta, True)
        if k != n_steps - 1:
            _native.check(lib.bvi_leapfrog_mid(blocks, nb, c, eps, ptr(z), ptr(p), ptr(theta), ptr(ll), ptr(gl),
                                               ptr(lp), ptr(g), st), 'bvi_leapfrog_mid')
        else:
            _native.check(lib.bvi_leapfrog_post(blocks, nb, c, eps, ptr(z), ptr(p), ptr(theta), ptr(ll), ptr(gl),
                                                ptr(lp), ptr(g), st), 'bvi_leapfrog_post')
    return torch.cat([z, p], -1)
