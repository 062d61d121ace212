"""Tuners (hoomd/tune/sorter.py): the particle sorter that keeps the arrays in cell order."""
from . import _hoomd_amd as _cpp


class ParticleSorter:
    """Reorders particles in memory every ``trigger`` steps (default 200 as hoomd/tune/sorter.py:44).

    The reference walks a Hilbert curve over a 256^3 grid; here the permutation is read off a cell
    list (cells in memory order), which gives the same locality for the neighbor gathers."""

    def __init__(self, trigger=200, grid=None):
        self.trigger = int(trigger)
        self.grid = grid
        self._cpp_obj = None

    def _attach(self, sim):
        self._cpp_obj = _cpp.ParticleSorter(sim._sysdef, self.trigger)
        if self.grid is not None:
            L = sim._sysdef.getParticleData().getBox().L
            self._cpp_obj.setWidth(max(L) / float(self.grid))

    def _detach(self):
        self._cpp_obj = None
