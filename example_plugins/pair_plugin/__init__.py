"""Example pair-potential plugin (example_plugins/pair_plugin/__init__.py of the reference): what a third-party pair
potential consists of -- one evaluator header, one HIP translation unit naming the kernel instantiation, one module file
with two export lines, this Python wrapper, and a Makefile that builds them outside the framework's tree."""
from . import pair

__all__ = ["pair"]
