"""Stochastic adjoint (placeholder module: filled in below)."""
from .sde import BaseSDE


class AdjointSDE(BaseSDE):
    pass
