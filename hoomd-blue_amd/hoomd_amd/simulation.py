"""Simulation, State and Operations (hoomd/simulation.py, hoomd/state.py, hoomd/operations.py)."""
import numpy as np

from . import _hoomd_amd as _cpp
from . import filter as _filter
from .box import Box
from .snapshot import Snapshot
from .tune import ParticleSorter


class State:
    """Particle data of a simulation (hoomd/state.py)."""

    def __init__(self, sim, snapshot, decomposition=None):
        self._sim = sim
        box = snapshot.configuration._box
        if box is None:
            raise RuntimeError("snapshot.configuration.box is not set")
        self._global_box = Box.from_box(box)
        self._ndim = box.dimensions
        cbox = _cpp.BoxDim(box.Lx, box.Ly, box.Lz, box.xy, box.xz, box.yz)
        self._decomposition = decomposition
        if decomposition is None:
            self._sysdef = _cpp.SystemDefinition(snapshot._to_dict(), cbox, self._ndim, sim.seed)
        else:
            self._sysdef = decomposition._make_sysdef(snapshot, cbox, self._ndim, sim.seed)
        self._pdata = self._sysdef.getParticleData()
        if snapshot.bonds.N or snapshot.bonds.types:
            # Under domain decomposition every rank keeps the whole (static) bond topology keyed by tag; the per-rank
            # table by local index is rebuilt from it after every migration (partners are owned or ghost particles).
            self._sysdef.setBonds(*snapshot._bonds_to_args())

    @property
    def bond_types(self):
        return list(self._sysdef.getBonds()["types"])

    @property
    def box(self):
        return Box.from_box(self._global_box)

    @property
    def N_particles(self):
        return self._pdata.N_global

    @property
    def particle_types(self):
        return list(self._pdata.types)

    def get_snapshot(self):
        """Global snapshot in tag order (gathered on every rank under domain decomposition)."""
        d = self._pdata.takeSnapshot()
        if self._decomposition is not None:
            d = self._decomposition._gather_snapshot(d)
        snap = Snapshot._from_dict(d, self._global_box)
        snap._set_bonds(self._sysdef.getBonds())
        return snap

    def set_snapshot(self, snapshot):
        """Overwrite the particle data; N and the type list must not change (hoomd/state.py set_snapshot)."""
        if snapshot.particles.N != self._pdata.N_global:
            raise RuntimeError("set_snapshot cannot change the number of particles")
        if list(snapshot.particles.types) != list(self._pdata.types):
            raise RuntimeError("Particle types must remain the same")
        if self._decomposition is not None:
            self._decomposition._scatter_snapshot(snapshot, self._pdata)
        else:
            self._pdata.initializeFromSnapshot(snapshot._to_dict())
        self._sim._state_changed()

    def thermalize_particle_momenta(self, filter, kT):
        """Gaussian velocities at kT with the net momentum removed (ParticleGroup::
        thermalizeParticleMomenta, hoomd/ParticleGroup.cc:690-780; NumPy stream, not Philox)."""
        snap = self.get_snapshot()
        tags = filter(self)
        idx = np.arange(snap.particles.N) if tags is None else np.asarray(tags, dtype=np.int64)
        rng = np.random.default_rng([self._sim.seed, self._sim.timestep, 0x7468])
        m = snap.particles.mass[idx]
        v = rng.normal(0.0, 1.0, (len(idx), 3)) * np.sqrt(kT / m)[:, None]
        if self._ndim == 2:
            v[:, 2] = 0
        p = (v * m[:, None]).sum(0) / max(len(idx), 1)
        v -= p[None, :] / m[:, None]
        snap.particles.velocity[idx] = v
        self.set_snapshot(snap)


class _TunerList(list):
    pass


class Operations:
    """Holds the integrator, tuners and computes of a simulation (hoomd/operations.py)."""

    def __init__(self, sim):
        self._sim = sim
        self._integrator = None
        self.tuners = _TunerList([ParticleSorter()])  # hoomd/operations.py:75
        self.computes = []
        self.writers = []  # hoomd/operations.py: objects with a trigger and _write(sim)
        self._scheduled = False

    @property
    def integrator(self):
        return self._integrator

    @integrator.setter
    def integrator(self, value):
        if self._integrator is not None and self._integrator._cpp_obj is not None:
            self._integrator._detach()
        self._integrator = value
        self._scheduled = False

    def __iadd__(self, op):
        self.add(op)
        return self

    def add(self, op):
        from .md.integrate import Integrator
        if isinstance(op, Integrator):
            self.integrator = op
        elif isinstance(op, ParticleSorter):
            self.tuners.append(op)
        elif hasattr(op, "_write") and hasattr(op, "trigger"):
            self.writers.append(op)
        else:
            self.computes.append(op)
        self._scheduled = False

    def _schedule(self):
        """Attach everything that is not attached yet (hoomd/operations.py:170-190)."""
        sim = self._sim
        if sim._state is None:
            raise RuntimeError("System not initialized yet")
        if self._integrator is not None and self._integrator._cpp_obj is None:
            self._integrator._attach(sim)
        for c in self.computes:
            if c._cpp_obj is None:
                c._attach(sim)
        sorter = None
        for t in self.tuners:
            if isinstance(t, ParticleSorter):
                if t._cpp_obj is None:
                    t._attach(sim)
                sorter = t
        sim._cpp_sys.setIntegrator(self._integrator._cpp_obj if self._integrator is not None else None)
        sim._cpp_sys.setSorter(sorter._cpp_obj if sorter is not None else None)
        self._scheduled = True


class Simulation:
    """``Simulation(device, seed)`` (hoomd/simulation.py)."""

    def __init__(self, device, seed=None):
        self.device = device
        self._seed = 0 if seed is None else int(seed) & 0xFFFF
        self._state = None
        self._cpp_sys = None
        self._init_timestep = 0
        self.operations = Operations(self)
        self.always_compute_pressure = False
        self._groups = {}
        self._nlists = []
        self._thermos = []
        self._tps = 0.0
        self._walltime = 0.0

    @property
    def seed(self):
        return self._seed

    @seed.setter
    def seed(self, v):
        self._seed = int(v) & 0xFFFF

    @property
    def state(self):
        return self._state

    @property
    def _sysdef(self):
        return self._state._sysdef

    @property
    def timestep(self):
        return self._cpp_sys.getCurrentTimeStep() if self._cpp_sys is not None else self._init_timestep

    @timestep.setter
    def timestep(self, step):
        if self._state is not None:
            raise RuntimeError("State must not be set to change timestep.")
        self._init_timestep = int(step)

    @property
    def tps(self):
        """Average time steps per second of the last run (hoomd/simulation.py:357)."""
        return self._tps

    @property
    def walltime(self):
        return self._walltime

    def create_state_from_snapshot(self, snapshot, domain_decomposition=None):
        """hoomd/simulation.py:222-300.  ``domain_decomposition``: a ``hoomd_amd.comm.DomainDecomposition``
        (one rank per GPU) or None for a single GPU."""
        if self._state is not None:
            raise RuntimeError("Cannot initialize more than once\n")
        self._state = State(self, snapshot, domain_decomposition)
        self._cpp_sys = _cpp.System(self._state._sysdef, self._init_timestep)
        if domain_decomposition is not None:
            domain_decomposition._attach(self)

    def create_state_from_gsd(self, filename, frame=-1, domain_decomposition=None):
        """hoomd/simulation.py:180-220: the state and the time step of one frame of a GSD file."""
        if self._state is not None:
            raise RuntimeError("Cannot initialize more than once\n")
        from .write.gsd import read_frame
        snapshot, step = read_frame(str(filename), int(frame))
        if self._init_timestep == 0:
            self._init_timestep = step
        self.create_state_from_snapshot(snapshot, domain_decomposition)

    # ---- helpers used by the operations when they attach ----
    def _get_group(self, flt):
        if flt not in self._groups:
            tags = flt(self._state)
            if tags is None:
                g = _cpp.ParticleGroup(self._sysdef)
            else:
                g = _cpp.ParticleGroup(self._sysdef, [int(t) for t in tags])
            self._groups[flt] = g
        return self._groups[flt]

    def _register_nlist(self, nlist):
        if nlist not in self._nlists:
            self._nlists.append(nlist)
        if self._state._decomposition is not None:
            self._state._decomposition._nlist_attached(nlist)

    def _register_thermo(self, thermo):
        self._thermos.append(thermo)
        if self._state._decomposition is not None:
            thermo.setAllreduce(self._state._decomposition._allreduce_sum_list)

    def _request_pressure(self):
        self.always_compute_pressure = True

    def _state_changed(self):
        for nl in self._nlists:
            if nl._cpp_obj is not None:
                nl._cpp_obj.forceUpdate()

    def run(self, steps, write_at_start=False):
        """Advance ``steps`` time steps (hoomd/simulation.py:485-565)."""
        if self._state is None:
            raise RuntimeError("Cannot run before state is set.")
        if steps < 0 or int(steps) != steps:
            raise ValueError("steps must be a non-negative integer")
        self.operations._schedule()  # attaches whatever was added since the last run
        integ = self.operations.integrator
        if integ is not None:
            for g in self._groups.values():
                integ._cpp_obj.updateGroupDOF(g)
        self._cpp_sys.setPressureFlag(bool(self.always_compute_pressure))
        if integ is None:
            if steps:
                raise RuntimeError("Cannot run without an integrator")
            return
        writers = list(self.operations.writers)
        if not writers:
            self._cpp_sys.run(int(steps))
            self._tps = self._cpp_sys.getLastTPS()
            self._walltime = self._cpp_sys.getLastWalltime()
            return
        # hoomd/System.cc:80-190: writers act on the state after a step, when their trigger fires at the new
        # time step (and before the first step with write_at_start); the run is cut at those steps
        if write_at_start:
            for w in writers:
                if w.trigger(self.timestep):
                    w._write(self)
        done, wall = 0, 0.0
        steps = int(steps)
        while done < steps:
            t0 = self.timestep
            k = next((j for j in range(1, steps - done + 1) if any(w.trigger(t0 + j) for w in writers)), steps - done)
            # only the first piece is a run's start (forced migration, forces of the initial state); writers read
            # the state between the pieces and change nothing
            self._cpp_sys.run(k, done == 0)
            wall += self._cpp_sys.getLastWalltime()
            done += k
            for w in writers:
                if w.trigger(self.timestep):
                    w._write(self)
        for w in writers:
            w.flush()
        self._walltime = wall
        self._tps = steps / wall if wall > 0 else 0.0
