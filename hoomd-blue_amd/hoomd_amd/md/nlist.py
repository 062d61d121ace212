"""Neighbor lists (hoomd/md/nlist.py): the cell-list based ``Cell``."""
import numpy as np

from .. import _hoomd_amd as _cpp
from ..error import DataAccessError
from ..operation import AutotunedObject


class NeighborList(AutotunedObject):
    """Base class (hoomd/md/nlist.py:84-323)."""

    def __init__(self, buffer, exclusions, rebuild_check_delay, check_dist, mesh):
        if mesh is not None:
            raise NotImplementedError("mesh-bond exclusions are outside this backend's scope")
        self._buffer = float(buffer)
        self._exclusions = tuple(exclusions)
        self._rebuild_check_delay = int(rebuild_check_delay)
        self._check_dist = bool(check_dist)
        self._cpp_obj = None
        self._sim = None

    def _prop(name, cpp_name, cast):
        def get(self):
            return getattr(self._cpp_obj, cpp_name) if self._cpp_obj is not None else getattr(self, "_" + name)

        def set_(self, v):
            setattr(self, "_" + name, cast(v))
            if self._cpp_obj is not None:
                setattr(self._cpp_obj, cpp_name, cast(v))

        return property(get, set_)

    buffer = _prop("buffer", "buffer", float)
    rebuild_check_delay = _prop("rebuild_check_delay", "rebuild_check_delay", int)
    check_dist = _prop("check_dist", "check_dist", bool)
    del _prop

    @property
    def exclusions(self):
        return self._exclusions

    @exclusions.setter
    def exclusions(self, value):
        value = tuple(value)
        for e in value:
            if e not in ("bond", "angle", "constraint", "dihedral", "special_pair", "body", "1-3", "1-4", "meshbond"):
                raise ValueError(f"unknown exclusion {e}")
        self._exclusions = value
        if self._cpp_obj is not None:
            self._apply_exclusions()

    def _apply_exclusions(self):
        # hoomd/md/NeighborList.cc:383-393: topology-derived exclusions.  Bonds are the only topology
        # this backend holds ('angle' etc. resolve to nothing); 'body' turns on the body filter.
        self._cpp_obj.clearExclusions()
        if "bond" in self._exclusions:
            self._cpp_obj.addExclusionsFromBonds()
        self._cpp_obj.setFilterBody("body" in self._exclusions)

    def add_exclusion(self, tag_i, tag_j):
        """Exclude one pair by tag (NeighborList::addExclusion, hoomd/md/NeighborList.cc:569)."""
        self._require_attached("add_exclusion")
        self._cpp_obj.addExclusion(int(tag_i), int(tag_j))

    def _kernel_parameters(self):
        # block sizes of the cell-list, neighbor-list and distance-check kernels (fixed for gfx950)
        return {"cell_list": (256,), "nlist_binned": (256,), "nlist_needs_update": (256,)}

    def _require_attached(self, name):
        if self._cpp_obj is None:
            raise DataAccessError(name)

    @property
    def shortest_rebuild(self):
        """Shortest period between builds (hoomd/md/nlist.py:307-315)."""
        self._require_attached("shortest_rebuild")
        return self._cpp_obj.getSmallestRebuild()

    @property
    def num_builds(self):
        """Number of builds since the last run() (hoomd/md/nlist.py:317-323)."""
        self._require_attached("num_builds")
        return self._cpp_obj.getNumUpdates()

    @property
    def pair_list(self):
        """(M, 2) array of tag pairs with tag_i < tag_j (hoomd/md/nlist.py:235-262)."""
        self._require_attached("pair_list")
        self._cpp_obj.compute(self._sim.timestep)
        p = np.asarray(self._cpp_obj.getPairList())
        if p.size:
            p = np.sort(p, axis=1)
        return p

    @property
    def local_pair_list(self):
        """(M, 2) array of local index pairs (hoomd/md/nlist.py:264-290)."""
        self._require_attached("local_pair_list")
        self._cpp_obj.compute(self._sim.timestep)
        return np.asarray(self._cpp_obj.getLocalPairList())

    @property
    def cpu_local_nlist_arrays(self):
        """Host copies of (head_list, n_neigh, nlist), the LocalNeighborListData role
        (hoomd/md/NeighborList.h:662-739)."""
        self._require_attached("cpu_local_nlist_arrays")
        self._cpp_obj.compute(self._sim.timestep)
        n = self._sim._sysdef.getParticleData().N
        return dict(head_list=np.asarray(self._cpp_obj.getHeadListArray(n)), n_neigh=np.asarray(self._cpp_obj.getNNeighArray(n)),
                    nlist=np.asarray(self._cpp_obj.getNListArray()), half_nlist=False)


class Cell(NeighborList):
    """Cell-list neighbor list (hoomd/md/nlist.py:326-418)."""

    def __init__(self, buffer, exclusions=('bond',), rebuild_check_delay=1, check_dist=True, deterministic=False, mesh=None):
        super().__init__(buffer, exclusions, rebuild_check_delay, check_dist, mesh)
        self._deterministic = bool(deterministic)

    @property
    def deterministic(self):
        return self._cpp_obj.deterministic if self._cpp_obj is not None else self._deterministic

    @deterministic.setter
    def deterministic(self, v):
        self._deterministic = bool(v)
        if self._cpp_obj is not None:
            self._cpp_obj.deterministic = bool(v)

    def _attach(self, sim):
        if self._cpp_obj is not None:
            if self._sim is not sim:
                raise RuntimeError("neighbor list is already attached to another simulation")
            return
        self._sim = sim
        self._cpp_obj = _cpp.NeighborListGPUBinned(sim._sysdef, self._buffer)  # hoomd/md/nlist.py:390-397
        self._cpp_obj.rebuild_check_delay = self._rebuild_check_delay
        self._cpp_obj.check_dist = self._check_dist
        self._cpp_obj.deterministic = self._deterministic
        self._apply_exclusions()
        sim._register_nlist(self)

    def _detach(self):
        self._cpp_obj = None
        self._sim = None

    @property
    def dimensions(self):
        """Cell-list dimensions (hoomd/md/nlist.py:399-408)."""
        self._require_attached("dimensions")
        return tuple(self._cpp_obj.getDim())

    @property
    def allocated_particles_per_cell(self):
        """Slots allocated per cell (hoomd/md/nlist.py:410-418)."""
        self._require_attached("allocated_particles_per_cell")
        return self._cpp_obj.getNmax()
