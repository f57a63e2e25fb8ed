"""fp64 restatement of the leapfrog recipe and of one HMC transition (TEST ORACLE).

Test infrastructure only -- see ``oracle/__init__.py``.

``solve`` follows bayes_vi/utils/symplectic_integrators.py:38-59 operation by
operation; its known answer is the harmonic-oscillator orbit of the
reference's README.md:571-591.  ``hmc_step`` restates what
``tfp.mcmc.HamiltonianMonteCarlo`` (reached through
bayes_vi/inference/mcmc/transition_kernels.py:122-128) does per transition,
from the published algorithm (recalled, TFP is not installed): momenta
p ~ N(0, I), L leapfrog steps on the *target's* gradient, Metropolis accept on
the energy difference, non-finite proposals rejected.  Momenta and uniforms are
injected so that a device implementation can be compared draw for draw.
"""
from __future__ import annotations

import numpy as np


def solve(grad_potential, grad_kinetic, initial_state, step_sizes, num_integration_steps):
    """Leapfrog on a separable Hamiltonian H(q,p) = U(q) + T(p).

    grad_potential / grad_kinetic take [..., d] and return the gradient of the
    (batch-summed) energy, i.e. per-sample gradients -- what ``tf.gradients``
    yields at symplectic_integrators.py:45,50,53,56.
    """
    x = np.asarray(initial_state, dtype=np.float64)
    if x.shape[-1] % 2 != 0:                                  # :39-42
        raise ValueError('Not a valid state, has to have even event dimension!; was {}'.format(x.shape))
    d = x.shape[-1] // 2
    q, p = x[..., :d], x[..., d:]                             # :40
    eps = np.asarray(step_sizes, dtype=np.float64)
    p = p - 0.5 * eps * grad_potential(q)                     # :45
    for n in range(num_integration_steps):                    # :48
        q = q + eps * grad_kinetic(p)                         # :50
        if n != num_integration_steps - 1:                    # :52
            p = p - eps * grad_potential(q)                   # :53
    p = p - 0.5 * eps * grad_potential(q)                     # :56
    return np.concatenate([q, p], -1)                         # :59


def hmc_trajectory(value_and_grad, q, p, step_size, num_leapfrog_steps):
    """L leapfrog steps on log-density ``value_and_grad(q) -> (lp[C], g[C,P])``.

    Sign convention of HMC (potential = -log pi): p += eps/2 * grad log pi.
    Returns q', p', lp', g'.
    """
    q = np.array(q, dtype=np.float64); p = np.array(p, dtype=np.float64)
    eps = np.asarray(step_size, dtype=np.float64)
    lp, g = value_and_grad(q)
    for _ in range(num_leapfrog_steps):
        p = p + 0.5 * eps * g
        q = q + eps * p
        lp, g = value_and_grad(q)
        p = p + 0.5 * eps * g
    return q, p, lp, g


def hmc_step(value_and_grad, q, momenta, log_uniform, step_size, num_leapfrog_steps):
    """One HMC transition for C chains with injected randomness.

    q[C,P], momenta[C,P] ~ N(0,1), log_uniform[C] = log U(0,1).
    Returns (q_next, is_accepted[C], log_accept_ratio[C], lp_next[C]).
    """
    q = np.asarray(q, dtype=np.float64)
    lp0, _ = value_and_grad(q)
    q1, p1, lp1, _ = hmc_trajectory(value_and_grad, q, momenta, step_size, num_leapfrog_steps)
    k0 = 0.5 * (np.asarray(momenta, dtype=np.float64) ** 2).sum(-1)
    k1 = 0.5 * (p1 ** 2).sum(-1)
    log_ratio = (lp1 - k1) - (lp0 - k0)
    log_ratio = np.where(np.isfinite(log_ratio), log_ratio, -np.inf)
    acc = np.asarray(log_uniform) < log_ratio
    q_next = np.where(acc[:, None], q1, q)
    return q_next, acc, log_ratio, np.where(acc, lp1, lp0)
