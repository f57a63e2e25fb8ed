"""MCMC inference (mirror of bayes_vi/inference/mcmc/__init__.py)."""
from .mcmc import MCMC  # noqa: F401
