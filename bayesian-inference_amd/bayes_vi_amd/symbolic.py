"""Recognising a likelihood callable without evaluating it on data.

The reference hands the user's `likelihood(**params[, features])` to
`JointDistributionNamedAutoBatched`, which calls it on real tensors at every
log-prob evaluation (bayes_vi/model/model.py:194-203).  Here the callable is
called ONCE with symbolic stand-ins for the parameters and the feature columns;
the distribution object it returns, with affine symbolic expressions as
arguments, is matched against the families the HIP kernels fuse
(iid Normal, Gaussian linear regression, univariate Gaussian mixture).  A
likelihood outside those families raises `UnsupportedLikelihood`: there is no
generic fallback path.

Supported algebra on symbols: +, -, unary -, * by python numbers, parameter *
feature products, `matvec(features, w)` / `features @ w` for a vector parameter,
and indexing `w[j]`, `features['name']`, `features[:, j]`.
"""
from __future__ import annotations

import numbers


class UnsupportedLikelihood(NotImplementedError):
    pass


class Affine:
    """sum_t coef_t * param_t * feature_t, where param_t / feature_t may be absent (=1).

    terms: list of (param_ref | None, feature_ref | None, coef) with
    param_ref = (name, index | None) and feature_ref = column key.
    """

    def __init__(self, terms):
        self.terms = list(terms)

    # ---- construction helpers ----
    @staticmethod
    def lift(v):
        if isinstance(v, Affine):
            return v
        if isinstance(v, ParamSym):
            return v.as_affine()
        if isinstance(v, FeatureColumn):
            return Affine([(None, v.key, 1.0)])
        if isinstance(v, numbers.Number):
            return Affine([(None, None, float(v))])
        raise UnsupportedLikelihood('cannot use %r in a likelihood expression' % (v,))

    def __add__(self, o):
        return Affine(self.terms + Affine.lift(o).terms)
    __radd__ = __add__

    def __neg__(self):
        return Affine([(p, f, -c) for p, f, c in self.terms])

    def __sub__(self, o):
        return self + (-Affine.lift(o))

    def __rsub__(self, o):
        return Affine.lift(o) + (-self)

    def __mul__(self, o):
        o = Affine.lift(o)
        out = []
        for p1, f1, c1 in self.terms:
            for p2, f2, c2 in o.terms:
                if (p1 is not None and p2 is not None) or (f1 is not None and f2 is not None):
                    raise UnsupportedLikelihood('only expressions linear in the parameters and in the features '
                                                'are fused (found a product of two parameters or two features)')
                out.append((p1 if p1 is not None else p2, f1 if f1 is not None else f2, c1 * c2))
        return Affine(out)
    __rmul__ = __mul__

    def __truediv__(self, o):
        if isinstance(o, numbers.Number):
            return self * (1.0 / o)
        raise UnsupportedLikelihood('division by a symbolic quantity is not fused')


class ParamSym:
    """Stand-in for one model parameter (scalar or vector)."""

    def __init__(self, name, size, index=None):
        self.name, self.size, self.index = name, size, index

    def as_affine(self):
        if self.size != 1 and self.index is None:
            raise UnsupportedLikelihood('vector parameter %s used as a scalar; index it or use matvec' % self.name)
        return Affine([((self.name, self.index), None, 1.0)])

    def __getitem__(self, j):
        if self.index is not None or not isinstance(j, int):
            raise UnsupportedLikelihood('parameters can only be indexed by one integer')
        return ParamSym(self.name, self.size, j)

    def _a(self):
        return self.as_affine()

    def __add__(self, o): return self._a() + o
    __radd__ = __add__
    def __sub__(self, o): return self._a() - o
    def __rsub__(self, o): return Affine.lift(o) - self._a()
    def __neg__(self): return -self._a()
    def __mul__(self, o): return self._a() * o
    __rmul__ = __mul__
    def __truediv__(self, o): return self._a() / o
    def __rmatmul__(self, feats):
        return matvec(feats, self)


class FeatureColumn:
    def __init__(self, key):
        self.key = key

    def _a(self): return Affine.lift(self)
    def __add__(self, o): return self._a() + o
    __radd__ = __add__
    def __sub__(self, o): return self._a() - o
    def __rsub__(self, o): return Affine.lift(o) - self._a()
    def __neg__(self): return -self._a()
    def __mul__(self, o): return self._a() * o
    __rmul__ = __mul__


class FeatureSet:
    """Stand-in for `features`: a dict of named columns or an [N, F] matrix."""

    def __init__(self, keys, is_dict):
        self.keys, self.is_dict = list(keys), is_dict

    def __getitem__(self, k):
        if self.is_dict:
            if k not in self.keys:
                raise KeyError(k)
            return FeatureColumn(k)
        if isinstance(k, tuple) and len(k) == 2 and k[0] == slice(None) and isinstance(k[1], int):
            return FeatureColumn(self.keys[k[1]])
        if k is Ellipsis or (isinstance(k, tuple) and k[0] is Ellipsis and isinstance(k[-1], int)):
            return FeatureColumn(self.keys[k[-1]]) if isinstance(k, tuple) else self
        raise UnsupportedLikelihood('features can be indexed as features[:, j] (tensor) or features[name] (dict)')

    def __matmul__(self, w):
        return matvec(self, w)

    def __mul__(self, o):
        # a single-column feature tensor used directly as a column (README.md:754-756 with format 'tensor')
        if len(self.keys) == 1:
            return FeatureColumn(self.keys[0]) * o
        raise UnsupportedLikelihood('multiply a parameter with one feature column, or use matvec(features, w)')
    __rmul__ = __mul__


def matvec(features, w):
    """features[N, F] . w[F] for a symbolic feature matrix and a vector parameter; with concrete tensors
    (posterior-predictive evaluation: features[N, F], w[..., F]) the plain product [..., N]."""
    import torch
    if isinstance(features, torch.Tensor) and isinstance(w, torch.Tensor):
        return w.to(features.dtype) @ features.T
    if not isinstance(features, FeatureSet) or not isinstance(w, ParamSym) or w.index is not None:
        raise UnsupportedLikelihood('matvec(features, w) needs the feature matrix and a whole vector parameter')
    if w.size != len(features.keys):
        raise ValueError('matvec: %d feature columns but parameter %s has %d entries'
                         % (len(features.keys), w.name, w.size))
    return Affine([((w.name, j if w.size > 1 else None), features.keys[j], 1.0) for j in range(w.size)])


class CholeskySolveIdentity:
    """(P P^T)^-1 for a whole [D, D] parameter P: what `linalg.cholesky_solve(P, linalg.eye(D))` returns on a
    symbolic parameter."""

    def __init__(self, param):
        self.param = param
        self.size = param.size


def is_plain_param(v, size=None):
    """v is exactly one whole parameter (optionally of the given size)."""
    if isinstance(v, ParamSym) and v.index is None:
        return size is None or v.size == size
    return False
