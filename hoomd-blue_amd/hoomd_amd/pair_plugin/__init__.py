"""Example pair-potential plugin (example_plugins/pair_plugin/__init__.py): shows what a third-party
evaluator needs -- an evaluator class in csrc/evaluators.cuh, one HIP translation unit that
instantiates the pair-force template for it (csrc/pair_force_example.hip), one export line in the host
module, and this Python wrapper."""
from . import pair

__all__ = ["pair"]
