"""Constraining bijectors on torch tensors (forward = constrain, inverse = unconstrain).

Mirrors the bijector semantics the reference relies on
(bayes_vi/model/model.py:51-55, 112-116): a `Model` holds one bijector per
parameter; the sampler runs in unconstrained space and evaluates
target(forward(z)) + forward_log_det_jacobian(z)
(bayes_vi/inference/mcmc/mcmc.py:153-156).  TFP's defaults are restated from its
published behaviour: Identity for real-valued priors, Softplus for positive
ones, SoftmaxCentered for the simplex, FillScaleTriL for Cholesky factors.

Every bijector acts on the LAST axis of a [..., n] tensor (parameters are kept
flat) and offers, besides forward/inverse/fldj, the two closed-form pieces the
hot loop needs without autograd:
    vjp(x, g_y)   = J(x)^T g_y          (chain rule of the log-density gradient)
    grad_fldj(x)  = d fldj / d x
All of forward/inverse/fldj are differentiable torch code, so autograd through
them (variational inference) works as well.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


class Bijector:
    name = 'bijector'

    def forward(self, x):
        raise NotImplementedError

    def inverse(self, y):
        raise NotImplementedError

    def forward_log_det_jacobian(self, x):
        raise NotImplementedError

    def inverse_log_det_jacobian(self, y):
        return -self.forward_log_det_jacobian(self.inverse(y))

    def vjp(self, x, g):
        raise NotImplementedError

    def grad_fldj(self, x):
        raise NotImplementedError

    def forward_event_size(self, n):
        return n

    def inverse_event_size(self, n):
        return n

    def __call__(self, x):
        return self.forward(x)

    def __repr__(self):
        return '%s()' % type(self).__name__


class Identity(Bijector):
    def forward(self, x): return x
    def inverse(self, y): return y
    def forward_log_det_jacobian(self, x): return x.new_zeros(x.shape[:-1])
    def vjp(self, x, g): return g
    def grad_fldj(self, x): return torch.zeros_like(x)


class Softplus(Bijector):
    def forward(self, x): return F.softplus(x)
    def inverse(self, y): return y + torch.log(-torch.expm1(-y))
    def forward_log_det_jacobian(self, x): return (-F.softplus(-x)).sum(-1)
    def vjp(self, x, g): return g * torch.sigmoid(x)
    def grad_fldj(self, x): return torch.sigmoid(-x)


class Exp(Bijector):
    def forward(self, x): return torch.exp(x)
    def inverse(self, y): return torch.log(y)
    def forward_log_det_jacobian(self, x): return x.sum(-1)
    def vjp(self, x, g): return g * torch.exp(x)
    def grad_fldj(self, x): return torch.ones_like(x)


class SoftmaxCentered(Bijector):
    """R^{K-1} -> K-simplex, y = softmax([x, 0]); fldj = 0.5 log K + sum_k log y_k."""

    def _logy(self, x):
        xp = torch.cat([x, x.new_zeros(x.shape[:-1] + (1,))], -1)
        return torch.log_softmax(xp, -1)

    def forward(self, x): return torch.exp(self._logy(x))
    def inverse(self, y):
        ly = torch.log(y)
        return ly[..., :-1] - ly[..., -1:]
    def forward_log_det_jacobian(self, x):
        ly = self._logy(x)
        return 0.5 * math.log(ly.shape[-1]) + ly.sum(-1)
    def vjp(self, x, g):
        y = self.forward(x)
        return (y * (g - (y * g).sum(-1, keepdim=True)))[..., :-1]
    def grad_fldj(self, x):
        y = self.forward(x)
        return 1.0 - y.shape[-1] * y[..., :-1]
    def forward_event_size(self, n): return n + 1
    def inverse_event_size(self, n): return n - 1


def fill_triangular_index(d):
    """Position in the lower triangle of a D x D matrix -> index into the packed
    vector, following TFP's fill_triangular rule (reshape(concat([x[D:], reverse(x)]), [D, D]),
    lower part kept): [1..6] -> [[4,0,0],[6,5,0],[3,2,1]]."""
    m = d * (d + 1) // 2
    idx = torch.arange(m)
    full = torch.cat([idx[d:], idx.flip(0)]).reshape(d, d)
    return full


class FillScaleTriL(Bijector):
    """Packed vector [D(D+1)/2] -> lower-triangular matrix flattened to [D*D], with
    softplus(.) + diag_shift on the diagonal (tfb.FillScaleTriL(Softplus(), diag_shift)).
    With `batch` = B the flat vector holds B such blocks back to back ([B, D(D+1)/2] ->
    [B, D, D] flattened): a parameter like scale_tril[K, D, D] of a mixture."""

    def __init__(self, d=None, diag_shift=1e-5, batch=1, diag_bijector=None):
        if diag_bijector is not None and not isinstance(diag_bijector, Softplus):
            raise NotImplementedError('FillScaleTriL: only diag_bijector=Softplus() is implemented')
        self.d, self.diag_shift, self.batch = d, diag_shift, batch
        if d is not None:
            self._build()

    def bind(self, event_shape):
        """TFP infers the matrix size from its input; written without `d` (tfb.FillScaleTriL(diag_bijector=...,
        diag_shift=...), multivariate-gaussian-test.ipynb cell 15) the size comes from the parameter's event shape
        [..., D, D] when the model is built."""
        if self.d is None:
            shape = tuple(event_shape)
            if len(shape) < 2 or shape[-1] != shape[-2]:
                raise ValueError('FillScaleTriL constrains [..., D, D] parameters; got event shape %s' % (shape,))
            self.d = int(shape[-1])
            self.batch = int(math.prod(shape[:-2])) if len(shape) > 2 else 1
            self._build()
        return self

    def _build(self):
        d, batch = self.d, self.batch
        full = fill_triangular_index(d)
        rows, cols = torch.tril_indices(d, d)
        m, dd = d * (d + 1) // 2, d * d
        src1, dst1, isd1 = full[rows, cols], rows * d + cols, rows == cols
        off_u = (torch.arange(batch) * m)[:, None]
        off_c = (torch.arange(batch) * dd)[:, None]
        self._src = (src1[None] + off_u).reshape(-1)     # packed index of every lower-tri entry
        self._dst = (dst1[None] + off_c).reshape(-1)     # flat position in the B*D*D output
        self._isdiag = isd1[None].expand(batch, -1).reshape(-1)

    def _to(self, dev):
        # index tensors are uploaded once per device (a per-call upload blocks and cannot be graph-captured)
        cache = self.__dict__.setdefault('_dev_cache', {})
        key = str(dev)
        if key not in cache:
            cache[key] = (self._src.to(dev), self._dst.to(dev), self._isdiag.to(dev))
        return cache[key]

    def forward(self, x):
        src, dst, isd = self._to(x.device)
        v = x[..., src]
        v = torch.where(isd, F.softplus(v) + self.diag_shift, v)
        out = x.new_zeros(x.shape[:-1] + (self.batch * self.d * self.d,))
        return out.index_copy(-1, dst, v)

    def inverse(self, y):
        src, dst, isd = self._to(y.device)
        v = y[..., dst]
        vd = (v - self.diag_shift).clamp_min(1e-30)
        v = torch.where(isd, vd + torch.log(-torch.expm1(-vd)), v)
        out = y.new_zeros(y.shape[:-1] + (self.batch * self.d * (self.d + 1) // 2,))
        return out.index_copy(-1, src, v)

    def forward_log_det_jacobian(self, x):
        src, _, isd = self._to(x.device)
        return (-F.softplus(-x[..., src[isd]])).sum(-1)

    def vjp(self, x, g):
        src, dst, isd = self._to(x.device)
        gv = g[..., dst]
        gv = torch.where(isd, gv * torch.sigmoid(x[..., src]), gv)
        return torch.zeros_like(x).index_copy(-1, src, gv)

    def grad_fldj(self, x):
        src, _, isd = self._to(x.device)
        out = torch.zeros_like(x)
        return out.index_copy(-1, src[isd], torch.sigmoid(-x[..., src[isd]]))

    def forward_event_size(self, n): return self.batch * self.d * self.d
    def inverse_event_size(self, n): return self.batch * self.d * (self.d + 1) // 2

    def __repr__(self):
        return 'FillScaleTriL(d=%d, diag_shift=%g, batch=%d)' % (self.d, self.diag_shift, self.batch)


class Blockwise(Bijector):
    """Applies one bijector per contiguous block of the flat parameter vector:
    the restatement of `Model.blockwise_constraining_bijector`
    (bayes_vi/model/model.py:155-161: split -> reshape -> constrain -> flatten -> concat)."""

    def __init__(self, bijectors, unconstrained_sizes):
        self.bijectors = list(bijectors)
        self.usizes = list(unconstrained_sizes)
        self.csizes = [b.forward_event_size(n) for b, n in zip(self.bijectors, self.usizes)]

    def _split(self, t, sizes):
        return torch.split(t, sizes, dim=-1)

    def forward(self, x):
        return torch.cat([b.forward(p) for b, p in zip(self.bijectors, self._split(x, self.usizes))], -1)

    def inverse(self, y):
        return torch.cat([b.inverse(p) for b, p in zip(self.bijectors, self._split(y, self.csizes))], -1)

    def forward_log_det_jacobian(self, x):
        return sum(b.forward_log_det_jacobian(p) for b, p in zip(self.bijectors, self._split(x, self.usizes)))

    def vjp(self, x, g):
        return torch.cat([b.vjp(p, q) for b, p, q in
                          zip(self.bijectors, self._split(x, self.usizes), self._split(g, self.csizes))], -1)

    def grad_fldj(self, x):
        return torch.cat([b.grad_fldj(p) for b, p in zip(self.bijectors, self._split(x, self.usizes))], -1)

    def forward_event_size(self, n): return sum(self.csizes)
    def inverse_event_size(self, n): return sum(self.usizes)
