"""Symplectic integrators (mirror of bayes_vi/utils/symplectic_integrators.py)."""
import torch


class LeapfrogIntegrator:
    """Leapfrog integrator for a separable Hamiltonian H(q, p) = U(q) + T(p).

    `solve` takes arbitrary differentiable energy callables, as the reference's does
    (symplectic_integrators.py:12-59, used by HamiltonianFlow, flow_bijectors.py:67-75):
    gradients come from torch autograd of the batch-summed energy (per-sample gradients for
    batch-independent energies, like tf.gradients).  It stays differentiable w.r.t. the
    state, the step sizes and the energies' parameters when gradients are enabled.
    An energy object may expose `.grad(x)` (closed-form gradient) to skip autograd.
    """

    def __init__(self):
        pass

    @staticmethod
    def _grad(energy_fn, x):
        if hasattr(energy_fn, 'grad'):
            return energy_fn.grad(x)
        track = torch.is_grad_enabled() and (x.requires_grad or any(
            p.requires_grad for p in getattr(energy_fn, 'parameters', lambda: [])()))
        with torch.enable_grad():
            xx = x if x.requires_grad else x.detach().requires_grad_(True)
            e = energy_fn(xx).sum()
            g, = torch.autograd.grad(e, xx, create_graph=track)
        return g if track else g.detach()

    def solve(self, potential_energy_fn, kinetic_energy_fn, initial_state, step_sizes, num_integration_steps):
        """initial_state[..., 2d] = [q | p] -> final state [..., 2d] after n leapfrog steps."""
        x = torch.as_tensor(initial_state)
        if x.shape[-1] % 2 != 0:
            raise ValueError('Not a valid state, has to have even event dimension!; was {}'.format(tuple(x.shape)))
        d = x.shape[-1] // 2
        q, p = x[..., :d], x[..., d:]
        eps = step_sizes
        p = p - 0.5 * eps * self._grad(potential_energy_fn, q)
        for n in range(num_integration_steps):
            q = q + eps * self._grad(kinetic_energy_fn, p)
            if n != num_integration_steps - 1:
                p = p - eps * self._grad(potential_energy_fn, q)
        p = p - 0.5 * eps * self._grad(potential_energy_fn, q)
        return torch.cat([q, p], -1)
