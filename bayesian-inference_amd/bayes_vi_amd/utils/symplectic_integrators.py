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

    Fast path: when the potential is the model's own (`fused_target.PotentialEnergy`, U = -log posterior), the
    kinetic energy is `fused_target.QuadraticKineticEnergy`, the step size is one number and no gradient of the
    result is asked for, the steps run on the device-resident leapfrog kernels around the fused data pass
    (`fused_target.leapfrog_on_device`): the integrator of HMC, with the same update order.
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

    @staticmethod
    def _device_route(potential_energy_fn, kinetic_energy_fn, x, step_sizes):
        from ..fused_target import PotentialEnergy, QuadraticKineticEnergy, leapfrog_on_device
        if not (isinstance(potential_energy_fn, PotentialEnergy) and isinstance(kinetic_energy_fn, QuadraticKineticEnergy)):
            return None
        post = potential_energy_fn.posterior
        scalar = isinstance(step_sizes, (int, float)) or (
            isinstance(step_sizes, torch.Tensor) and step_sizes.dim() == 0 and not step_sizes.requires_grad)
        if not (scalar and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
                and not (x.requires_grad and torch.is_grad_enabled())
                and x.shape[-1] == 2 * post.n_unconstrained and post.device_blocks() is not None):
            return None
        return lambda n: leapfrog_on_device(post, x, float(step_sizes), n)

    def solve(self, potential_energy_fn, kinetic_energy_fn, initial_state, step_sizes, num_integration_steps):
        """initial_state[..., 2d] = [q | p] -> final state [..., 2d] after n leapfrog steps."""
        x = torch.as_tensor(initial_state)
        if x.shape[-1] % 2 != 0:
            raise ValueError('Not a valid state, has to have even event dimension!; was {}'.format(tuple(x.shape)))
        fast = self._device_route(potential_energy_fn, kinetic_energy_fn, x, step_sizes)
        if fast is not None:
            return fast(num_integration_steps)
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
