"""Utility functions and classes (the names of bayes_vi/utils/__init__.py:16-30)."""
from collections import OrderedDict
from functools import reduce

__all__ = ['datasets', 'symplectic_integrators', 'compose', 'to_ordered_dict']


def compose(fns):
    """One callable that feeds its arguments to fns[0] and then unpacks each result tuple into the next function
    of the list; the last function's result is returned.  `MCMC.fit` chains the trace functions of nested kernels
    with it, outermost first (bayes_vi/inference/mcmc/mcmc.py:172)."""
    steps = tuple(fns)
    if not steps:
        raise ValueError('compose needs at least one function')

    def composed(*args):
        return reduce(lambda result, fn: fn(*result), steps[1:], steps[0](*args))
    return composed


def to_ordered_dict(param_names, state):
    """Pairs parameter names with the state parts given in the same order."""
    return OrderedDict(zip(param_names, state))
