"""Neighbor lists (hoomd/md/nlist.py): the cell-list based ``Cell``."""
import numpy as np

from .. import _hoomd_amd as _cpp
from ..error import DataAccessError
from ..operation import AutotunedObject


class DeviceArrayView:
    """Read-only view of one of the neighbor list's device buffers (role of hoomd.data.array.HOOMDGPUArray,
    hoomd/data/array.py): carries ``__cuda_array_interface__`` (version 2: shape, typestr, data pointer), so
    ``torch.as_tensor(view, device="cuda")`` or ``cupy.asarray(view)`` wrap the same memory without a copy.  The
    buffers are the neighbor list's own and must not be written to (``read_only`` says so; the interface's own
    read-only flag stays clear because torch refuses to wrap memory that carries it).  Valid only inside the ``with``
    block that handed it out."""

    def __init__(self, owner, ptr, shape, typestr):
        self._owner = owner
        self._ptr, self._shape, self._typestr = int(ptr), tuple(int(v) for v in shape), typestr

    @property
    def __cuda_array_interface__(self):
        self._owner._check()
        return {"shape": self._shape, "typestr": self._typestr, "data": (self._ptr, False), "version": 2}

    shape = property(lambda self: self._shape)
    dtype = property(lambda self: np.dtype(self._typestr))
    read_only = True

    def __len__(self):
        return self._shape[0]


class NeighborListLocalAccessGPU:
    """``with nlist.gpu_local_nlist_arrays as data:`` -- the list's buffers on the GPU (hoomd/md/data/local_access_gpu.py,
    C++ side LocalNeighborListDataDevice, hoomd/md/NeighborList.h:662-739).  The neighbors of local particle i are
    ``nlist[head_list[i] : head_list[i] + n_neigh[i]]`` (local indices; ghosts have indices >= N).  Rank-local and
    read-only; the views die with the block (the next rebuild may move or rewrite the buffers)."""

    def __init__(self, nlist):
        self._nlist = nlist
        self._entered = False
        self._views = None

    def _check(self):
        if not self._entered:
            raise RuntimeError("neighbor-list device arrays are only valid inside the `with` block of gpu_local_nlist_arrays")

    def __enter__(self):
        nl = self._nlist
        if nl._in_context_manager:
            raise RuntimeError("Cannot enter gpu_local_nlist_arrays context manager inside another local_nlist_arrays context manager")
        cpp = nl._cpp_obj
        n = nl._sim._sysdef.getParticleData().N
        size = int(cpp.nlist_size)           # expands index rows from the tile rows when the last build wrote those
        _cpp.synchronize()                   # consumers on other streams see finished buffers
        self._views = dict(head_list=DeviceArrayView(self, cpp.head_list_ptr, (n,), "<u8"),
                           n_neigh=DeviceArrayView(self, cpp.n_neigh_ptr, (n,), "<u4"),
                           nlist=DeviceArrayView(self, cpp.nlist_ptr, (size,), "<u4"))
        nl._in_context_manager = True
        self._entered = True
        return self

    def __exit__(self, *exc):
        self._entered = False
        self._nlist._in_context_manager = False
        self._views = None
        return False

    def _get(self, name):
        self._check()
        return self._views[name]

    head_list = property(lambda self: self._get("head_list"))
    n_neigh = property(lambda self: self._get("n_neigh"))
    nlist = property(lambda self: self._get("nlist"))

    @property
    def half_nlist(self):
        self._check()
        return False  # the GPU path stores the full list (hoomd/md/NeighborListGPUBinned.cc:51-54)


class NeighborList(AutotunedObject):
    """Base class (hoomd/md/nlist.py:84-323)."""
    _in_context_manager = False

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

    @property
    def gpu_local_nlist_arrays(self):
        """Zero-copy, read-only device views of (head_list, n_neigh, nlist) for use in a ``with`` block
        (hoomd/md/nlist.py:187-233)."""
        self._require_attached("gpu_local_nlist_arrays")
        if self._in_context_manager:
            raise RuntimeError("Cannot enter gpu_local_nlist_arrays context manager inside another local_nlist_arrays context manager")
        self._cpp_obj.compute(self._sim.timestep)
        return NeighborListLocalAccessGPU(self)


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
