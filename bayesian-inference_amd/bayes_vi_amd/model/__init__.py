"""Package providing the `Model` class (mirror of bayes_vi/model/__init__.py)."""
from .model import Model  # noqa: F401
