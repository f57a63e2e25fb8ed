"""Inference algorithms (mirror of bayes_vi/inference/__init__.py)."""
from .inference import Inference  # noqa: F401
