"""``ExamplePair`` (example_plugins/pair_plugin/pair.py:15-28): U(r) = k/2 (sigma - r)^2 for r < r_cut."""
from hoomd_amd.md import pair as _pair

# the plugin's own extension module, built by the Makefile next to the sources
from . import _pair_plugin


class ExamplePair(_pair.Pair):
    """Example pair potential; ``params``: dict(k, sigma) per type pair."""

    _ext_module = _pair_plugin
    _cpp_class_name = "PotentialPairExample"
    _accepted_modes = ("none", "shift", "xplor")

    def _convert_params(self, value):
        missing = {"k", "sigma"} - set(value)
        if missing:
            raise ValueError(f"ExamplePair params need {sorted(missing)}")
        return dict(k=float(value["k"]), sigma=float(value["sigma"]))
