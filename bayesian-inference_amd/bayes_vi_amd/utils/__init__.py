"""Utility functions and classes (mirror of bayes_vi/utils/__init__.py:16-30)."""
import collections
import functools

__all__ = ['datasets', 'symplectic_integrators', 'compose', 'to_ordered_dict']


def compose(fns):
    """Left-to-right composition: the result tuple of each function is splatted into the next
    (how the reference chains its trace functions, bayes_vi/utils/__init__.py:16-25)."""
    def composition(*args, fns_):
        res = fns_[0](*args)
        for f in fns_[1:]:
            res = f(*res)
        return res
    return functools.partial(composition, fns_=fns)


def to_ordered_dict(param_names, state):
    """List of state parts -> OrderedDict keyed by parameter name (utils/__init__.py:28-30)."""
    return collections.OrderedDict([(k, v) for k, v in zip(param_names, state)])
