"""fp32 torch-CPU restatement of how the reference's TFP path evaluates the
log-likelihood and its gradient (TEST ORACLE and timed CPU baseline).

Test infrastructure only -- see ``oracle/__init__.py``.  This is *not* TFP: it
is the same mathematics evaluated the way ``JointDistributionNamedAutoBatched
.log_prob`` evaluates it behind the closure of
bayes_vi/inference/mcmc/mcmc.py:57-62 -- elementwise ops on materialised
``[chains, rows]`` float32 tensors followed by a sum over rows, with gradients
by reverse-mode autodiff (what ``tfp.math.value_and_gradient`` does inside the
HMC/NUTS kernels of transition_kernels.py:122-128, 238-246).  ``bench.py``
times it on the host cores as the ``cpu_baseline`` (kind "port").

A ``[C, N]`` float32 temporary at C=1024, N=1e7 is 41 GB, so every function
takes ``chunk`` (rows per slice) and accumulates value and gradient over
slices; with ``chunk=None`` the whole batch is evaluated at once exactly as the
reference would.
"""
from __future__ import annotations

import math

import torch

LN2PI = math.log(2.0 * math.pi)


def _normal_log_prob(y, loc, scale):
    # tfd.Normal.log_prob: -0.5 * ((y - loc) / scale)^2 - 0.5 log(2 pi) - log(scale)
    z = (y - loc) / scale
    return -0.5 * z * z - 0.5 * LN2PI - torch.log(scale)


def _ll_normal_iid(theta, y):
    return _normal_log_prob(y[None, :], theta[:, 0:1], theta[:, 1:2]).sum(-1)


def _ll_linreg(theta, X, y):
    d = X.shape[1]
    mu = theta[:, 0:1] + theta[:, 1:1 + d] @ X.T                # [C, N]
    return _normal_log_prob(y[None, :], mu, theta[:, 1 + d:2 + d]).sum(-1)


def _ll_gmm1d(theta, y, k):
    w, mu, sc = theta[:, :k], theta[:, k:2 * k], theta[:, 2 * k:]
    logw = torch.log_softmax(torch.log(w), -1)                  # Categorical(probs=w)
    comp = _normal_log_prob(y[None, None, :], mu[:, :, None], sc[:, :, None])   # [C,K,N]
    return torch.logsumexp(comp + logw[:, :, None], dim=1).sum(-1)


def _value_and_grad(fn, theta, data, chunk):
    theta = theta.detach().to(torch.float32)
    n = data[-1].shape[0]
    chunk = n if chunk is None else chunk
    val = torch.zeros(theta.shape[0]); grad = torch.zeros_like(theta)
    for s in range(0, n, chunk):
        t = theta.clone().requires_grad_(True)
        v = fn(t, *[a[s:s + chunk] for a in data])
        g, = torch.autograd.grad(v.sum(), t)
        val += v.detach(); grad += g
    return val, grad


def normal_iid(y, theta, chunk=None):
    return _value_and_grad(_ll_normal_iid, theta, (y,), chunk)


def linreg(X, y, theta, chunk=None):
    return _value_and_grad(_ll_linreg, theta, (X, y), chunk)


def gmm1d(y, theta, k, chunk=None):
    return _value_and_grad(lambda t, yy: _ll_gmm1d(t, yy, k), theta, (y,), chunk)
