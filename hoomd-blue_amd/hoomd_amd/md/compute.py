"""Thermodynamic properties of a group (hoomd/md/compute.py:24-300, the translational subset)."""
from .. import _hoomd_amd as _cpp
from ..error import DataAccessError


class ThermodynamicProperties:
    def __init__(self, filter):
        self.filter = filter
        self._cpp_obj = None
        self._sim = None

    def _attach(self, sim):
        self._sim = sim
        self._group = sim._get_group(self.filter)
        self._cpp_obj = _cpp.ComputeThermoGPU(sim._sysdef, self._group)
        sim._register_thermo(self._cpp_obj)

    def _detach(self):
        self._cpp_obj = None
        self._sim = None

    def _get(self, name):
        if self._cpp_obj is None:
            raise DataAccessError(name)
        pdata = self._sim._sysdef.getParticleData()
        flags = pdata.getFlags()
        if name == "pressure":
            # the virial is only accumulated on steps that ask for it (hoomd/System.cc:216-239);
            # asking here recomputes the forces of this step with the pressure flag set
            pdata.setFlags(flags | 1)
        integ = self._sim.operations.integrator
        if integ is not None and integ._cpp_obj is not None:
            integ._cpp_obj.updateGroupDOF(self._group)
            integ._cpp_obj.computeNetForce(self._sim.timestep)
        self._cpp_obj.forceCompute(self._sim.timestep)
        value = getattr(self._cpp_obj, name)
        pdata.setFlags(flags)
        return value

    kinetic_energy = property(lambda self: self._get("kinetic_energy"))
    potential_energy = property(lambda self: self._get("potential_energy"))
    kinetic_temperature = property(lambda self: self._get("kinetic_temperature"))
    pressure = property(lambda self: self._get("pressure"))
    degrees_of_freedom = property(lambda self: self._get("degrees_of_freedom"))
    translational_degrees_of_freedom = degrees_of_freedom

    @property
    def num_particles(self):
        return self._group.getNumMembersGlobal()

    @property
    def volume(self):
        return self._sim.state.box.volume


#: name used by the reference (hoomd/md/compute.py:24)
ThermodynamicQuantities = ThermodynamicProperties
