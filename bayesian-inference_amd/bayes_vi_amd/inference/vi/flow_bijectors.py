"""Trainable bijectors for normalizing flows (mirror of bayes_vi/inference/vi/flow_bijectors.py).

The energies of `HamiltonianFlow` default to small MLPs (tanh hidden layers, softplus output,
flow_bijectors.py:144-153): O(samples x hidden^2) work with no pass over the dataset, so they are
plain torch modules differentiated by autograd; only the target log-density underneath the
ELBO goes through the fused kernels.
"""
import torch
from torch import nn

from ...bijectors import Bijector
from ...utils.symplectic_integrators import LeapfrogIntegrator


def _mlp(in_dim, hidden_layers, out_dim, hidden_act, out_act):
    layers, d = [], in_dim
    for n in hidden_layers:
        layers += [nn.Linear(d, n), hidden_act()]
        d = n
    layers.append(nn.Linear(d, out_dim))
    if out_act is not None:
        layers.append(out_act())
    return nn.Sequential(*layers)


class _ReLU6(nn.Module):
    def forward(self, x):
        return torch.clamp(x, 0.0, 6.0)


def make_energy_fn(event_dims, hidden_layers=None):
    """MLP energy: tanh hidden layers [32, 32] by default, softplus scalar output (flow_bijectors.py:144-153)."""
    return _mlp(event_dims, hidden_layers or [32, 32], 1, nn.Tanh, nn.Softplus)


def make_scale_fn(event_ndims, hidden_layers=None):
    """MLP for a diagonal scale: relu6 hidden [32], softplus output (flow_bijectors.py:156-165)."""
    return _mlp(event_ndims, hidden_layers or [32], event_ndims, _ReLU6, nn.Softplus)


def make_shift_fn(event_ndims, hidden_layers=None):
    """MLP for a location: relu6 hidden [32], linear output (flow_bijectors.py:168-177)."""
    return _mlp(event_ndims, hidden_layers or [32], event_ndims, _ReLU6, None)


class Chain(Bijector, nn.Module):
    """Composition of trainable flow bijectors with tfb.Chain's order: `forward` applies the LAST bijector of the
    list first (the reference stacks Hamiltonian flows this way, linear-regression-example.ipynb cell 13)."""

    def __init__(self, bijectors, name='chain'):
        nn.Module.__init__(self)
        self.bijectors = nn.ModuleList(bijectors) if all(isinstance(b, nn.Module) for b in bijectors) \
            else list(bijectors)
        self.name = name

    def forward(self, x):
        for b in reversed(list(self.bijectors)):
            x = b.forward(x)
        return x

    def inverse(self, y):
        for b in self.bijectors:
            y = b.inverse(y)
        return y

    def forward_log_det_jacobian(self, x):
        total = x.new_zeros(x.shape[:-1])
        for b in reversed(list(self.bijectors)):
            total = total + b.forward_log_det_jacobian(x)
            x = b.forward(x)
        return total


class HamiltonianFlow(Bijector, nn.Module):
    """Volume-preserving flow: n leapfrog steps of a (trainable) separable Hamiltonian on the
    augmented state [q | p]; inverse = the same integration with negated step sizes;
    log|det J| = 0 (flow_bijectors.py:15-85)."""

    def __init__(self, event_dims, potential_energy_fn=None, kinetic_energy_fn=None,
                 symplectic_integrator=None, step_sizes=0.1, num_integration_steps=5,
                 hidden_layers=None, validate_args=False, name='hamiltonian_flow'):
        nn.Module.__init__(self)
        self.potential_energy_fn = potential_energy_fn if potential_energy_fn is not None \
            else make_energy_fn(event_dims, hidden_layers)
        self.kinetic_energy_fn = kinetic_energy_fn if kinetic_energy_fn is not None \
            else make_energy_fn(event_dims, hidden_layers)
        self.symplectic_integrator = symplectic_integrator if symplectic_integrator is not None else LeapfrogIntegrator()
        self.num_integration_steps = num_integration_steps
        self.step_sizes = step_sizes        # float, tensor, or nn.Parameter (trainable)
        self.name = name

    def forward(self, x):
        return self.symplectic_integrator.solve(self.potential_energy_fn, self.kinetic_energy_fn, x,
                                                self.step_sizes, self.num_integration_steps)

    def inverse(self, y):
        return self.symplectic_integrator.solve(self.potential_energy_fn, self.kinetic_energy_fn, y,
                                                -self.step_sizes, self.num_integration_steps)

    def forward_log_det_jacobian(self, x):
        return x.new_zeros(x.shape[:-1])

    def inverse_log_det_jacobian(self, y):
        return y.new_zeros(y.shape[:-1])


class AffineFlow(Bijector, nn.Module):
    """Trainable affine flow y = L x + b with L lower triangular, softplus diagonal
    (flow_bijectors.py:88-141; initial L = 0.5 I, b ~ N(0, 1))."""

    def __init__(self, event_dims, validate_args=False, name='affine_flow'):
        nn.Module.__init__(self)
        self.d = event_dims
        self.shift = nn.Parameter(torch.randn(event_dims))
        raw = torch.zeros(event_dims, event_dims)
        raw.diagonal().fill_(float(torch.log(torch.expm1(torch.tensor(0.5)))))
        self.raw_tril = nn.Parameter(raw)
        self.name = name

    def _tril(self):
        L = torch.tril(self.raw_tril, -1)
        return L + torch.diag(torch.nn.functional.softplus(self.raw_tril.diagonal()))

    def forward(self, x):
        return x @ self._tril().T + self.shift

    def inverse(self, y):
        return torch.linalg.solve_triangular(self._tril(), (y - self.shift).unsqueeze(-1), upper=False).squeeze(-1)

    def forward_log_det_jacobian(self, x):
        return torch.log(torch.nn.functional.softplus(self.raw_tril.diagonal())).sum().expand(x.shape[:-1])
