"""Integration methods (hoomd/md/methods/methods.py, hoomd/md/methods/thermostats.py)."""
from .. import _hoomd_amd as _cpp
from .. import filter as _filter
from ..operation import AutotunedObject


class Thermostat:
    _cpp_obj = None

    def __init__(self, kT):
        self._kT = float(kT)

    @property
    def kT(self):
        return self._kT

    @kT.setter
    def kT(self, v):
        self._kT = float(v)
        if self._cpp_obj is not None:
            self._cpp_obj.kT = float(v)


class Berendsen(Thermostat):
    """Berendsen velocity rescaling (hoomd/md/methods/thermostats.py:228-290)."""

    def __init__(self, kT, tau):
        super().__init__(kT)
        self.tau = float(tau)

    def _attach(self, sim, group, thermo):
        self._cpp_obj = _cpp.BerendsenThermostat(self._kT, group, thermo, sim._sysdef, self.tau)


class Bussi(Thermostat):
    """Bussi-Donadio-Parrinello stochastic rescaling (hoomd/md/methods/thermostats.py:149-226).
    The random stream differs from the reference's Philox stream; the sampled ensemble does not."""

    def _attach(self, sim, group, thermo):
        self._cpp_obj = _cpp.BussiThermostat(self._kT, group, thermo, sim._sysdef)


class thermostats:  # namespace mirror of hoomd.md.methods.thermostats
    Berendsen = Berendsen
    Bussi = Bussi
    Thermostat = Thermostat


class Method(AutotunedObject):
    _cpp_obj = None

    def _kernel_parameters(self):
        return {"step_one": (256,), "step_two": (256,)}


class ConstantVolume(Method):
    """Constant volume dynamics, NVE without and NVT with a thermostat
    (hoomd/md/methods/methods.py:85-174)."""

    def __init__(self, filter, thermostat=None):
        if not isinstance(filter, _filter.ParticleFilter):
            raise TypeError("filter must be a hoomd_amd.filter.ParticleFilter")
        self.filter = filter
        self._thermostat = thermostat
        self._max_disp = None
        self._sim = None

    @property
    def thermostat(self):
        return self._thermostat

    @thermostat.setter
    def thermostat(self, t):
        self._thermostat = t
        if self._cpp_obj is not None:
            self._attach_thermostat()

    @property
    def maximum_displacement(self):
        return self._max_disp

    @maximum_displacement.setter
    def maximum_displacement(self, v):
        self._max_disp = None if v is None else float(v)
        if self._cpp_obj is not None:
            self._cpp_obj.maximum_displacement = -1.0 if v is None else float(v)

    def _attach_thermostat(self):
        if self._thermostat is None:
            self._cpp_obj.setThermostat(None)
        else:
            self._thermostat._attach(self._sim, self._group, self._thermo)
            self._cpp_obj.setThermostat(self._thermostat._cpp_obj)

    def _attach(self, sim):
        # hoomd/md/methods/methods.py:155-174
        self._sim = sim
        self._group = sim._get_group(self.filter)
        self._thermo = _cpp.ComputeThermoGPU(sim._sysdef, self._group)
        sim._register_thermo(self._thermo)
        self._cpp_obj = _cpp.TwoStepConstantVolumeGPU(sim._sysdef, self._group, None)
        self._attach_thermostat()
        if self._max_disp is not None:
            self._cpp_obj.maximum_displacement = self._max_disp

    def _detach(self):
        self._cpp_obj = None
        self._sim = None
