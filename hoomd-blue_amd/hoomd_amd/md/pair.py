"""Pair potentials (hoomd/md/pair/pair.py): the ``Pair`` base and ``LJ``."""
import numpy as np

from .. import _hoomd_amd as _cpp
from ..error import DataAccessError, MutabilityError
from ..typeparam import PairTypeParameter
from ..operation import AutotunedObject
from .nlist import NeighborList

_MODES = {"none": 0, "shift": 1, "xplor": 2}


def _pos_float(v):
    v = float(v)
    if v < 0:
        raise ValueError("value must be non-negative")
    return v


class Force(AutotunedObject):
    """Per-particle results of a force compute (hoomd/md/force.py:60-215)."""

    _cpp_obj = None
    _sim = None

    def _ready(self, name):
        if self._cpp_obj is None:
            raise DataAccessError(name)
        self._cpp_obj.compute(self._sim.timestep)

    @property
    def forces(self):
        """(N, 3) forces in tag order."""
        self._ready("forces")
        return np.asarray(self._cpp_obj.getForcesAndEnergies())[:, :3].copy()

    @property
    def energies(self):
        """(N,) per-particle energies in tag order."""
        self._ready("energies")
        return np.asarray(self._cpp_obj.getForcesAndEnergies())[:, 3].copy()

    @property
    def energy(self):
        """Total potential energy of this force."""
        self._ready("energy")
        return self._cpp_obj.calcEnergySum()

    @property
    def virials(self):
        """(N, 6) virial contributions xx, xy, xz, yy, yz, zz (needs the pressure flag, as in
        hoomd/md/force.py:175-196: computed when a pressure consumer is attached)."""
        if self._cpp_obj is None:
            raise DataAccessError("virials")
        pdata = self._sim._sysdef.getParticleData()
        flags = pdata.getFlags()
        pdata.setFlags(flags | 1)  # pressure_tensor: forces are recomputed with the virial
        self._cpp_obj.compute(self._sim.timestep)
        v = np.asarray(self._cpp_obj.getVirials())
        pdata.setFlags(flags)
        return v

    @property
    def additional_energy(self):
        self._ready("additional_energy")
        return self._cpp_obj.getExternalEnergy()

    @property
    def additional_virial(self):
        self._ready("additional_virial")
        return np.array([self._cpp_obj.getExternalVirial(i) for i in range(6)])


class Pair(Force):
    """Base of the isotropic pair potentials (hoomd/md/pair/pair.py:18-160)."""

    _cpp_class_name = None
    _ext_module = _cpp
    _accepted_modes = ("none", "shift", "xplor")

    def __init__(self, nlist, default_r_cut=None, default_r_on=0.0, mode="none"):
        if not isinstance(nlist, NeighborList):
            raise TypeError("nlist must be a hoomd_amd.md.nlist.NeighborList")
        self._nlist = nlist
        self.r_cut = PairTypeParameter("r_cut", _pos_float, None if default_r_cut is None else float(default_r_cut))
        self.r_on = PairTypeParameter("r_on", _pos_float, float(default_r_on))
        self.params = PairTypeParameter("params", self._convert_params)
        self._mode = None
        self.mode = mode

    def _convert_params(self, value):
        raise NotImplementedError

    @property
    def nlist(self):
        return self._nlist

    @nlist.setter
    def nlist(self, value):
        if self._cpp_obj is not None:
            raise MutabilityError("nlist")
        self._nlist = value

    @property
    def mode(self):
        return self._mode

    @mode.setter
    def mode(self, value):
        if value not in self._accepted_modes:
            raise ValueError(f"mode must be one of {self._accepted_modes}")
        self._mode = value
        if self._cpp_obj is not None:
            self._cpp_obj.mode = _MODES[value]

    def compute_energy(self, tags1, tags2):
        """U = sum over i in tags1, j in tags2 of V_ij(r) (hoomd/md/pair/pair.py:90-124); the sets are assumed
        disjoint and free of duplicates, as in the reference."""
        if self._cpp_obj is None:
            raise DataAccessError("compute_energy")
        t1 = [int(t) for t in np.asarray(tags1).ravel()]
        t2 = [int(t) for t in np.asarray(tags2).ravel()]
        return float(self._cpp_obj.computeEnergyBetweenSets(t1, t2))

    def _kernel_parameters(self):
        # (block size, lanes per particle); 0 lanes = the library's choice for the list at hand
        return {"pair_force": (256, int(self._cpp_obj.threads_per_particle))}

    def _set_kernel_parameter(self, name, values):
        if len(values) != 2 or values[0] != 256 or values[1] not in (0, 1, 2, 4, 8, 16, 32, 64):
            raise ValueError("pair_force takes (256, lanes per particle) with lanes a power of two up to 64, or 0")
        self._cpp_obj.threads_per_particle = values[1]

    def _attach(self, sim):
        """hoomd/md/pair/pair.py:126-144: attach the neighbor list, force the full storage mode the
        GPU kernels need, construct <_cpp_class_name>GPU, then push the type parameters."""
        self._sim = sim
        self._nlist._attach(sim)
        self._nlist._cpp_obj.setStorageMode(_cpp.NeighborListGPUBinned.full)
        cls = getattr(self._ext_module, self._cpp_class_name + "GPU")
        self._cpp_obj = cls(sim._sysdef, self._nlist._cpp_obj)
        self._cpp_obj.mode = _MODES[self._mode]
        types = sim._sysdef.getParticleData().types
        tid = {t: i for i, t in enumerate(types)}

        def key(pair):
            return (tid[pair[0]], tid[pair[1]])

        self.params._attach(types, lambda p, v: self._cpp_obj.setParams(key(p), v), lambda p: self._cpp_obj.getParams(key(p)))
        self.r_cut._attach(types, lambda p, v: self._cpp_obj.setRCut(key(p), v), lambda p: self._cpp_obj.getRCut(key(p)))
        self.r_on._attach(types, lambda p, v: self._cpp_obj.setROn(key(p), v), lambda p: self._cpp_obj.getROn(key(p)))

    def _detach(self):
        for tp in (self.params, self.r_cut, self.r_on):
            tp._detach()
        self._cpp_obj = None
        self._sim = None


class LJ(Pair):
    r"""Lennard-Jones pair force (hoomd/md/pair/pair.py:163-227):
    U(r) = 4 epsilon [ (sigma/r)^12 - (sigma/r)^6 ].  ``params``: dict(epsilon, sigma) per type pair."""

    _cpp_class_name = "PotentialPairLJ"

    def __init__(self, nlist, default_r_cut=None, default_r_on=0.0, mode="none", tail_correction=False):
        super().__init__(nlist, default_r_cut, default_r_on, mode)
        self._tail_correction = bool(tail_correction)

    def _convert_params(self, value):
        missing = {"epsilon", "sigma"} - set(value)
        if missing:
            raise ValueError(f"LJ params need {sorted(missing)}")
        return dict(epsilon=float(value["epsilon"]), sigma=float(value["sigma"]))

    @property
    def tail_correction(self):
        return self._tail_correction

    @tail_correction.setter
    def tail_correction(self, v):
        self._tail_correction = bool(v)
        if self._cpp_obj is not None:
            self._cpp_obj.tail_correction = bool(v)

    def _attach(self, sim):
        # hoomd/md/pair/pair.py:219-227 and hoomd/md/pytest/test_potential.py:1165-1187
        if self._tail_correction and self._mode != "none":
            raise RuntimeError('Pair potential shift mode must be "none" to apply tail corrections.')
        super()._attach(sim)
        self._cpp_obj.tail_correction = self._tail_correction
