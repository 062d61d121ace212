"""``ExamplePair`` (example_plugins/pair_plugin/pair.py:15-28): U(r) = k/2 (sigma - r)^2 for r < r_cut."""
from ..md import pair as _pair


class ExamplePair(_pair.Pair):
    """Example pair potential; ``params``: dict(k, sigma) per type pair."""

    _cpp_class_name = "PotentialPairExample"
    _accepted_modes = ("none", "shift", "xplor")

    def _convert_params(self, value):
        missing = {"k", "sigma"} - set(value)
        if missing:
            raise ValueError(f"ExamplePair params need {sorted(missing)}")
        return dict(k=float(value["k"]), sigma=float(value["sigma"]))
