"""The MD integrator (hoomd/md/integrate.py:141-326)."""
from .. import _hoomd_amd as _cpp
from ..operation import AutotunedObject


class _SyncedList(list):
    """List that notifies its owner when it changes (role of hoomd/data/syncedlist.py)."""

    def __init__(self, owner, items=()):
        super().__init__(items)
        self._owner = owner

    def _changed(self):
        self._owner._lists_changed()

    def append(self, x):
        super().append(x)
        self._changed()

    def extend(self, xs):
        super().extend(xs)
        self._changed()

    def remove(self, x):
        super().remove(x)
        self._changed()

    def clear(self):
        super().clear()
        self._changed()

    def __setitem__(self, i, x):
        super().__setitem__(i, x)
        self._changed()

    def __delitem__(self, i):
        super().__delitem__(i)
        self._changed()


class Integrator(AutotunedObject):
    """``Integrator(dt, methods=[...], forces=[...])``; constraints, rigid bodies, rotational degrees
    of freedom and half-step hooks of the reference are outside this backend's scope and rejected."""

    def __init__(self, dt, integrate_rotational_dof=False, forces=None, constraints=None, methods=None, rigid=None,
                 half_step_hook=None):
        if integrate_rotational_dof or constraints or rigid is not None or half_step_hook is not None:
            raise NotImplementedError("only translational point-particle dynamics are implemented")
        self._dt = float(dt)
        self._cpp_obj = None
        self._sim = None
        self._forces = _SyncedList(self, forces or [])
        self._methods = _SyncedList(self, methods or [])

    @property
    def dt(self):
        return self._dt

    @dt.setter
    def dt(self, v):
        self._dt = float(v)
        if self._cpp_obj is not None:
            self._cpp_obj.dt = self._dt

    @property
    def forces(self):
        return self._forces

    @forces.setter
    def forces(self, value):
        self._forces = _SyncedList(self, value)
        self._lists_changed()

    @property
    def methods(self):
        return self._methods

    @methods.setter
    def methods(self, value):
        self._methods = _SyncedList(self, value)
        self._lists_changed()

    integrate_rotational_dof = False

    def _lists_changed(self):
        if self._cpp_obj is not None:
            self._sync()

    def _sync(self):
        for f in self._forces:
            if f._cpp_obj is None:
                f._attach(self._sim)
        for m in self._methods:
            if m._cpp_obj is None:
                m._attach(self._sim)
        self._cpp_obj.setForces([f._cpp_obj for f in self._forces])
        self._cpp_obj.setMethods([m._cpp_obj for m in self._methods])
        for g in self._sim._groups.values():
            self._cpp_obj.updateGroupDOF(g)

    def _attach(self, sim):
        # hoomd/md/integrate.py:301-307
        self._sim = sim
        self._cpp_obj = _cpp.IntegratorTwoStep(sim._sysdef, self._dt)
        self._sync()

    def _detach(self):
        for f in self._forces:
            f._detach()
        for m in self._methods:
            m._detach()
        self._cpp_obj = None
        self._sim = None

    @property
    def linear_momentum(self):
        """Total linear momentum (hoomd/md/integrate.py:309-318)."""
        return tuple(self._cpp_obj.computeLinearMomentum())
