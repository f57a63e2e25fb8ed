"""The handful of `tf.linalg` functions the reference's likelihood callables use
(exploratory_experiments/multivariate-gaussian-test.ipynb cell 15: `cholesky_solve`, `eye`), on torch tensors and on
the symbolic stand-ins a likelihood is recognised with (`bayes_vi_amd.symbolic`)."""
from __future__ import annotations

import torch

from . import symbolic as sym


class Eye:
    """tf.linalg.eye(d) as written inside a likelihood callable."""

    def __init__(self, d):
        self.d = int(d)

    def tensor(self, like):
        return torch.eye(self.d, dtype=like.dtype, device=like.device)


def eye(num_rows):
    return Eye(num_rows)


def cholesky_solve(chol, rhs):
    """tf.linalg.cholesky_solve(chol, rhs): X with (chol chol^T) X = rhs.  With a symbolic parameter and `eye` as
    right-hand side it returns the description `CholeskySolveIdentity` (the full inverse (chol chol^T)^-1); with
    tensors, the solution [..., D, M] (only the lower triangle of `chol` is read)."""
    if isinstance(chol, sym.ParamSym):
        d = int(round(chol.size ** 0.5))
        if chol.index is not None or d * d != chol.size or not isinstance(rhs, Eye) or rhs.d != d:
            raise sym.UnsupportedLikelihood('cholesky_solve is fused for a whole [D, D] parameter and eye(D) only')
        return sym.CholeskySolveIdentity(chol)
    chol = torch.as_tensor(chol)
    b = rhs.tensor(chol) if isinstance(rhs, Eye) else torch.as_tensor(rhs, dtype=chol.dtype, device=chol.device)
    return torch.cholesky_solve(b.expand(chol.shape[:-2] + b.shape[-2:]), torch.tril(chol))
