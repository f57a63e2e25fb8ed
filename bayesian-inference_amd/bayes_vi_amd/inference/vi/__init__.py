"""Variational inference (mirror of bayes_vi/inference/vi/__init__.py)."""
from .vi import VI  # noqa: F401
