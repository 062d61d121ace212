"""Molecular dynamics operations (hoomd/md/__init__.py): the short-range pair path."""
from . import bond, compute, methods, nlist, pair
from .integrate import Integrator

__all__ = ["Integrator", "bond", "compute", "methods", "nlist", "pair"]
