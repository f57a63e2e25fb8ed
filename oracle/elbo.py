"""fp64 restatement of one optimisation step of the reference's ``VI.fit`` (TEST ORACLE).

Test infrastructure only -- see ``oracle/__init__.py``.  TFP cannot run here, so what pins this
module is what the reference itself holds for the VI path:
  * mean-field ADVI (``advi_step`` with ``Adam``): run with the experiments' settings (Adam 5e-3,
    10 000 steps, sample_size 50) on data standardised to the printed sample statistics, it
    reproduces for all ten datasets of both experiments the fitted surrogate's mean / sd per
    parameter (*/tables/vi_summary_statistics/vi_summary_statistics.txt, rows "meanfield advi") and
    the final average loss left in the notebooks' progress-bar outputs (cell 22 of
    univariate-gaussian-example.ipynb and linear-regression-example.ipynb; e.g. 200.060, 422.092) --
    i.e. the absolute value of the Monte-Carlo variational loss, constants, priors and Jacobian
    terms included, to a few hundredths (tests/test_oracle_pins.py::test_advi_meanfield_*_table);
  * full-rank ADVI and the flow surrogate: no reference output singles them out (the tables' flow
    rows come from random network initialisations); they are pinned against the mean-field code
    path they share (loss assembly, masked mean, clip, Adam) and, for the gradients, against fp64
    autograd and finite differences (tests/test_oracle_elbo.py).

What the reference computes per step
------------------------------------
``VI.fit`` (bayes_vi/inference/vi/vi.py:109-122) minimises

    loss = tfp.vi.monte_carlo_variational_loss(target_log_prob, surrogate.distribution,
               discrepancy_fn=safe_discrepancy_fn, use_reparameterization=True, sample_size=S)

i.e. with reparameterised draws q_s of the surrogate (published TFP 0.12 behaviour, recalled):
    logu_s = target_log_prob(q_s) - surrogate.log_prob(q_s),   d_s = discrepancy(logu_s) (= -logu_s, kl_reverse)
    loss   = mean of d_s over the FINITE entries only          (safe_discrepancy_fn, vi.py:55-58)
then differentiates it w.r.t. every trainable variable, clips each variable's gradient to norm 1
on its own (``tf.clip_by_norm(grad, 1.0)``, vi.py:70) and applies ``tf.optimizers.Adam`` (vi.py:72).

The surrogate lives in CONSTRAINED space (surrogate_posteriors.py:120-123, 131-134, 203-205):
a base density in unconstrained space pushed through ``model.blockwise_constraining_bijector``,
so with z the unconstrained draw and theta = f(z):
    surrogate.log_prob(theta) = log q_u(z) - forward_log_det_jacobian(z).
For augmented flows (``extra_ndims`` > 0) the target gains the lift density of the extra
coordinates a given the unconstrained parameters (surrogate_posteriors.py:79-88):
    target(theta, a) = log pi(theta) + lift(z).log_prob(a).

Two implementations live here:
  * numpy, closed-form gradients: mean-field and full-rank ADVI (surrogate_posteriors.py:113-134);
  * torch float64 + autograd: any surrogate, including ``NormalizingFlow`` over chained
    ``HamiltonianFlow`` bijectors with MLP energies (flow_bijectors.py:15-85, 144-177) and an MLP
    lift distribution; its leapfrog follows bayes_vi/utils/symplectic_integrators.py:38-59.
The noise of the reparameterised sampler is always an argument (injected), never drawn here.
"""
from __future__ import annotations

import math

import numpy as np

from . import closed_form as cf

LN2PI = math.log(2.0 * math.pi)
FULL_RANK_DIAG_SHIFT = 1e-5        # tfb.FillScaleTriL(diag_shift=1e-5), surrogate_posteriors.py:128


def softplus(x):
    return np.logaddexp(0.0, np.asarray(x, np.float64))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))


def softplus_inverse(y):
    y = np.asarray(y, np.float64)
    return y + np.log(-np.expm1(-y))


# ------------------------------------------------------------------------------------------
# pieces shared by both implementations
# ------------------------------------------------------------------------------------------
def blockwise_fldj(spec: cf.ModelSpec, z):
    """forward_log_det_jacobian of model.blockwise_constraining_bijector (model.py:155-161) at z[S,Pu]."""
    z = np.asarray(z, np.float64)
    out = np.zeros(z.shape[0]); o = 0
    for b in spec.blocks:
        n = spec._usize(b)
        out = out + b['bijector'].fldj(z[:, o:o + n]); o += n
    return out


def masked_mean(d):
    """Mean over the finite entries of d[S] (vi.py:55-58 followed by the reduce_mean of the Monte-Carlo
    expectation).  Returns (mean, finite mask, number of finite entries)."""
    d = np.asarray(d, np.float64)
    finite = np.isfinite(d)
    n = int(finite.sum())
    return (np.where(finite, d, 0.0).sum() / n if n else np.nan), finite, n


def clip_by_norm(g, clip_norm=1.0):
    """tf.clip_by_norm on one variable's gradient (vi.py:70): g * clip_norm / max(||g||_2, clip_norm)."""
    g = np.asarray(g, np.float64)
    return g * clip_norm / max(float(np.sqrt((g * g).sum())), clip_norm)


class Adam:
    """tf.optimizers.Adam as VI.fit applies it (vi.py:72, 77; published Keras update, recalled):
        m <- b1 m + (1 - b1) g;  v <- b2 v + (1 - b2) g^2
        lr_t = lr sqrt(1 - b2^t) / (1 - b1^t);  x <- x - lr_t m / (sqrt(v) + epsilon),  epsilon = 1e-7."""

    def __init__(self, params, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta_1, beta_2, epsilon
        self.t = 0
        self.m = {k: np.zeros_like(np.asarray(v, np.float64)) for k, v in params.items()}
        self.v = {k: np.zeros_like(np.asarray(v, np.float64)) for k, v in params.items()}

    def step(self, params, grads):
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        out = {}
        for k, x in params.items():
            g = np.asarray(grads[k], np.float64)
            self.m[k] = self.b1 * self.m[k] + (1.0 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1.0 - self.b2) * g * g
            out[k] = np.asarray(x, np.float64) - lr_t * self.m[k] / (np.sqrt(self.v[k]) + self.eps)
        return out


# ------------------------------------------------------------------------------------------
# ADVI in numpy with closed-form gradients
# ------------------------------------------------------------------------------------------
def advi_scale_tril(raw_tril):
    """The full-rank scale factor from its unconstrained variable (surrogate_posteriors.py:126-130):
    strictly-lower entries as they are, diagonal softplus(.) + 1e-5.  The variable is held as a [Pu, Pu] matrix
    whose upper triangle is unused (TFP keeps the same numbers as a vector of length Pu(Pu+1)/2: norms, and
    therefore clipping and the element-wise Adam update, are the same)."""
    raw = np.asarray(raw_tril, np.float64)
    return np.tril(raw, -1) + np.diag(softplus(np.diag(raw)) + FULL_RANK_DIAG_SHIFT)


def advi_sample_and_log_prob(spec: cf.ModelSpec, params, eps, mean_field=True):
    """Reparameterised draws of the ADVI surrogate and their log density (surrogate_posteriors.py:113-134).
    params: {'loc': [Pu], 'raw_scale': [Pu]} (mean field; scale = softplus(raw_scale), initial scale 1) or
            {'loc': [Pu], 'raw_tril': [Pu, Pu]} (full rank);   eps[S, Pu] standard-normal noise.
    Returns (z[S,Pu], theta[S,P], logq[S])."""
    eps = np.asarray(eps, np.float64); loc = np.asarray(params['loc'], np.float64)
    if mean_field:
        sc = softplus(params['raw_scale'])
        z = loc + sc * eps
        log_base = (-0.5 * eps * eps - 0.5 * LN2PI - np.log(sc)).sum(-1)
    else:
        L = advi_scale_tril(params['raw_tril'])
        z = loc + eps @ L.T
        log_base = (-0.5 * eps * eps - 0.5 * LN2PI).sum(-1) - np.log(np.diag(L)).sum()
    theta = spec.constrain(z)
    return z, theta, log_base - blockwise_fldj(spec, z)


def advi_loss_and_grads(spec: cf.ModelSpec, params, eps, mean_field=True):
    """loss (vi.py:109-116 with kl_reverse) and its gradient w.r.t. the surrogate's variables, closed form.

    d_s = log q(theta_s) - log pi(theta_s) = log q_u(z_s) - [log pi(f(z_s)) + fldj(z_s)]; the bracket is the
    unconstrained log posterior, whose z-gradient the model oracle has, so
        mean field:  d d_s/d loc = -g_s,   d d_s/d scale = -1/scale - g_s eps_s,  raw_scale via sigmoid
        full rank :  d d_s/d L   = -diag(1/L_dd) - tril(g_s eps_s^T),             diagonal of raw_tril via sigmoid
    averaged over the finite d_s only."""
    eps = np.asarray(eps, np.float64)
    z, theta, logq = advi_sample_and_log_prob(spec, params, eps, mean_field)
    lp_u, gz = spec.log_post_unconstrained(z)                    # = log pi(theta) + fldj(z)
    lp_c = lp_u - blockwise_fldj(spec, z)
    d = logq - lp_c
    # vi.py:55-58 drops the samples whose discrepancy is not finite.  A sample with a finite log density and a
    # non-finite gradient (a scale underflowing towards 0) would pass that test and make every variable NaN; the
    # restatement, like the product, counts a sample only if its log density AND its gradient are finite.
    d = np.where(np.isfinite(gz).all(-1), d, np.nan)
    loss, finite, n = masked_mean(d)
    w = finite.astype(np.float64)[:, None] / max(n, 1)
    gz = np.where(finite[:, None], gz, 0.0)
    grads = {'loc': -(w * gz).sum(0)}
    if mean_field:
        sc = softplus(params['raw_scale'])
        g_sc = -(w * (1.0 / sc + gz * eps)).sum(0)
        grads['raw_scale'] = g_sc * sigmoid(params['raw_scale'])
    else:
        L = advi_scale_tril(params['raw_tril'])
        g_L = -np.tril(np.einsum('s,si,sj->ij', w[:, 0], gz, eps)) - np.diag(w.sum() / np.diag(L))
        raw = np.asarray(params['raw_tril'], np.float64)
        g_raw = np.tril(g_L, -1) + np.diag(np.diag(g_L) * sigmoid(np.diag(raw)))
        grads['raw_tril'] = g_raw
    return loss, grads, dict(z=z, theta=theta, logq=logq, log_target=lp_c, finite=finite)


def advi_step(spec, params, eps, opt: Adam, mean_field=True):
    """One ``optimizer_step`` of vi.py:64-73.  Returns (loss, clipped gradients, new variables)."""
    loss, grads, _ = advi_loss_and_grads(spec, params, eps, mean_field)
    clipped = {k: clip_by_norm(g) for k, g in grads.items()}
    return loss, clipped, opt.step(params, clipped)


# ------------------------------------------------------------------------------------------
# any surrogate in torch float64 with autograd (CPU)
# ------------------------------------------------------------------------------------------
def _torch():
    import torch
    return torch


def t_log_posterior(family, data, blocks):
    """Constrained-space log pi(theta[S,P]) -> [S] in torch float64 for the families 'normal_iid' (data = (y,))
    and 'linreg' (data = (X, y)); priors as in ``cf.ModelSpec`` blocks (normal / halfnormal).  The same
    densities as ``closed_form`` (README.md:114-120, 747-756), written again on torch tensors so that autograd
    can differentiate a whole ELBO."""
    torch = _torch()
    data = [torch.as_tensor(np.asarray(a, np.float64)) for a in data]

    def log_prior(theta):
        lp = 0.0; o = 0
        for b in blocks:
            x = theta[:, o:o + b['size']]; o += b['size']
            kind = b['prior'][0]
            if kind == 'normal':
                loc, sc = float(b['prior'][1]), float(b['prior'][2])
                lp = lp + (-0.5 * ((x - loc) / sc) ** 2 - 0.5 * LN2PI - math.log(sc)).sum(-1)
            elif kind == 'halfnormal':
                sc = float(b['prior'][1])
                lp = lp + (0.5 * math.log(2.0 / math.pi) - math.log(sc) - 0.5 * (x / sc) ** 2).sum(-1)
            else:
                raise ValueError(kind)
        return lp

    def fn(theta):
        if family == 'normal_iid':
            y, = data
            mu, sc = theta[:, 0:1], theta[:, 1:2]
            r = y[None, :] - mu
        elif family == 'linreg':
            X, y = data
            d = X.shape[1]
            sc = theta[:, d + 1:d + 2]
            r = y[None, :] - theta[:, 0:1] - theta[:, 1:1 + d] @ X.T
        else:
            raise ValueError(family)
        n = r.shape[1]
        ll = (-0.5 * (r / sc) ** 2).sum(-1) - n * torch.log(sc[:, 0]) - 0.5 * n * LN2PI
        return ll + log_prior(theta)
    return fn


def t_constrain(blocks, z):
    """(theta, fldj) of the block-wise constraining bijector on a torch tensor z[S,Pu] (Identity / Softplus / Exp)."""
    torch = _torch()
    parts, fldj, o = [], 0.0, 0
    for b in blocks:
        bij = b['bijector']; n = b['size']
        x = z[:, o:o + n]; o += n
        if type(bij) is cf.Identity:
            parts.append(x)
        elif type(bij) is cf.Softplus:
            parts.append(torch.nn.functional.softplus(x)); fldj = fldj - torch.nn.functional.softplus(-x).sum(-1)
        elif type(bij) is cf.Exp:
            parts.append(torch.exp(x)); fldj = fldj + x.sum(-1)
        else:
            raise ValueError('bijector %s is not restated on torch tensors' % type(bij).__name__)
    return torch.cat(parts, -1), fldj


def t_mlp(x, layers, hidden_act, out_act):
    """tfk.Sequential of Dense layers (flow_bijectors.py:144-177).  layers: list of (kernel[in, out], bias[out])."""
    torch = _torch()
    acts = {'tanh': torch.tanh, 'relu6': lambda v: torch.clamp(v, 0.0, 6.0),
            'softplus': torch.nn.functional.softplus, None: lambda v: v}
    for i, (k, b) in enumerate(layers):
        x = x @ k + b
        x = acts[hidden_act](x) if i < len(layers) - 1 else acts[out_act](x)
    return x


def t_leapfrog(potential, kinetic, x, eps, n_steps):
    """bayes_vi/utils/symplectic_integrators.py:38-59 on a torch tensor, differentiable: the energies' gradients are
    those of the batch-summed energy (tf.gradients at :45, :50, :53, :56)."""
    torch = _torch()

    def grad(fn, v):
        if not v.requires_grad:
            v = v.detach().requires_grad_(True)
        g, = torch.autograd.grad(fn(v).sum(), v, create_graph=True)
        return g
    d = x.shape[-1] // 2
    q, p = x[..., :d], x[..., d:]                                  # :40
    p = p - 0.5 * eps * grad(potential, q)                         # :45
    for k in range(n_steps):                                       # :48
        q = q + eps * grad(kinetic, p)                             # :50
        if k != n_steps - 1:                                       # :52
            p = p - eps * grad(potential, q)                       # :53
    p = p - 0.5 * eps * grad(potential, q)                         # :56
    return torch.cat([q, p], -1)                                   # :59


class FlowSurrogate:
    """NormalizingFlow(model, flow_bijector=Chain([HamiltonianFlow, ...]), extra_ndims=Pu,
    posterior_lift_distribution=...) of surrogate_posteriors.py:153-212 with the experiment's pieces
    (linear_regression_experiment/linear-regression-example.ipynb cell 13).

    variables: dict name -> float64 array.  Per flow i (applied in the order tfb.Chain applies them: LAST first):
        'flow{i}.step'               scalar step size (tf.Variable(0.1))
        'flow{i}.U.{j}.k/.b', 'flow{i}.T.{j}.k/.b'   Dense kernels [in,out] / biases of the two energy MLPs
    and, when lift_layers > 0: 'lift.shift.{j}.k/.b', 'lift.scale.{j}.k/.b' (relu6 hidden, linear / softplus output;
    otherwise the lift is N(0, I), surrogate_posteriors.py:208-212)."""

    def __init__(self, blocks, n_flows, n_steps, n_energy_layers, n_lift_layers=0):
        self.blocks, self.n_flows, self.n_steps = blocks, n_flows, n_steps
        self.n_energy_layers, self.n_lift_layers = n_energy_layers, n_lift_layers
        self.pu = sum(b['size'] for b in blocks)      # Identity / Softplus / Exp blocks only: Pu == P

    def _layers(self, v, prefix, n):
        return [(v['%s.%d.k' % (prefix, j)], v['%s.%d.b' % (prefix, j)]) for j in range(n)]

    def loss(self, v, log_post, noise):
        """v: dict of torch float64 leaf tensors; noise[S, 2 Pu] base draw.  Returns (loss, per-sample pieces)."""
        torch = _torch()
        x = torch.as_tensor(np.asarray(noise, np.float64))
        log_base = (-0.5 * x * x - 0.5 * LN2PI).sum(-1)           # MultivariateNormalDiag(0, 1), :186-188
        u = x
        for i in reversed(range(self.n_flows)):                    # tfb.Chain: the last bijector acts first
            U = lambda q, i=i: t_mlp(q, self._layers(v, 'flow%d.U' % i, self.n_energy_layers), 'tanh', 'softplus')
            T = lambda p, i=i: t_mlp(p, self._layers(v, 'flow%d.T' % i, self.n_energy_layers), 'tanh', 'softplus')
            u = t_leapfrog(U, T, u, v['flow%d.step' % i], self.n_steps)     # ldj = 0, flow_bijectors.py:77-81
        z, a = u[:, :self.pu], u[:, self.pu:]
        theta, fldj = t_constrain(self.blocks, z)
        logq = log_base - fldj
        if self.n_lift_layers:
            shift = t_mlp(z, self._layers(v, 'lift.shift', self.n_lift_layers), 'relu6', None)
            scale = t_mlp(z, self._layers(v, 'lift.scale', self.n_lift_layers), 'relu6', 'softplus')
            lift = (-0.5 * ((a - shift) / scale) ** 2 - 0.5 * LN2PI - torch.log(scale)).sum(-1)
        else:
            lift = (-0.5 * a * a - 0.5 * LN2PI).sum(-1)
        target = log_post(theta) + lift                            # surrogate_posteriors.py:85-88
        d = logq - target                                          # kl_reverse(logu) = -logu
        finite = torch.isfinite(d)
        loss = torch.where(finite, d, torch.zeros_like(d)).sum() / finite.sum()
        return loss, dict(theta=theta, logq=logq, target=target)


def t_loss_and_grads(loss_fn, variables):
    """Evaluate loss_fn(dict of float64 leaf tensors) and its gradients by autograd -> (loss, {name: grad})."""
    torch = _torch()
    leaves = {k: torch.tensor(np.asarray(a, np.float64), requires_grad=True) for k, a in variables.items()}
    out = loss_fn(leaves)
    loss = out[0] if isinstance(out, tuple) else out
    names = list(leaves)
    gs = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    grads = {k: (np.zeros_like(np.asarray(variables[k], np.float64)) if g is None else g.numpy())
             for k, g in zip(names, gs)}
    return float(loss.detach()), grads


def t_advi_loss(spec_blocks, log_post, mean_field=True):
    """The ADVI loss as a function of torch leaves {'loc', 'raw_scale' | 'raw_tril'} and injected noise: the
    autograd twin of ``advi_loss_and_grads`` (used to pin its closed-form gradients)."""
    torch = _torch()

    def fn(v, eps):
        eps = torch.as_tensor(np.asarray(eps, np.float64))
        if mean_field:
            sc = torch.nn.functional.softplus(v['raw_scale'])
            z = v['loc'] + sc * eps
            log_base = (-0.5 * eps * eps - 0.5 * LN2PI - torch.log(sc)).sum(-1)
        else:
            raw = v['raw_tril']
            L = torch.tril(raw, -1) + torch.diag(torch.nn.functional.softplus(raw.diagonal()) + FULL_RANK_DIAG_SHIFT)
            z = v['loc'] + eps @ L.T
            log_base = (-0.5 * eps * eps - 0.5 * LN2PI).sum(-1) - torch.log(L.diagonal()).sum()
        theta, fldj = t_constrain(spec_blocks, z)
        d = (log_base - fldj) - log_post(theta)
        finite = torch.isfinite(d)
        return torch.where(finite, d, torch.zeros_like(d)).sum() / finite.sum()
    return fn
