"""Writers (hoomd/write/__init__.py): GSD trajectories."""
from .gsd import GSD

__all__ = ["GSD"]
