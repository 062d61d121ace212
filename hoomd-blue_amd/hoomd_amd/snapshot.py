"""In-memory system snapshot (hoomd/snapshot.py, hoomd/SnapshotSystemData.h)."""
import numpy as np

from .box import Box


class _Configuration:
    def __init__(self):
        self._box = None

    @property
    def box(self):
        return None if self._box is None else [self._box.Lx, self._box.Ly, self._box.Lz, self._box.xy, self._box.xz, self._box.yz]

    @box.setter
    def box(self, value):
        self._box = Box.from_box(value)

    @property
    def dimensions(self):
        return self._box.dimensions if self._box is not None else 3


class _Particles:
    _FIELDS = {
        "position": (np.float64, 3, 0.0), "velocity": (np.float64, 3, 0.0), "acceleration": (np.float64, 3, 0.0),
        "image": (np.int32, 3, 0), "typeid": (np.uint32, 0, 0), "body": (np.uint32, 0, 0xFFFFFFFF),
        "mass": (np.float64, 0, 1.0), "charge": (np.float64, 0, 0.0), "diameter": (np.float64, 0, 1.0),
    }

    def __init__(self):
        self._N = 0
        self.types = []
        self._alloc()

    def _alloc(self):
        for name, (dt, k, fill) in self._FIELDS.items():
            shape = (self._N, k) if k else (self._N,)
            setattr(self, name, np.full(shape, fill, dtype=dt))

    @property
    def N(self):
        return self._N

    @N.setter
    def N(self, n):
        """Resizing keeps existing rows (hoomd/ParticleData.cc SnapshotParticleData::resize)."""
        n = int(n)
        old = {name: getattr(self, name) for name in self._FIELDS}
        keep = min(n, self._N)
        self._N = n
        self._alloc()
        for name, a in old.items():
            getattr(self, name)[:keep] = a[:keep]


class _Bonds:
    """``snap.bonds.{N, types, typeid, group}`` (hoomd/snapshot.py, BondData::Snapshot)."""

    def __init__(self):
        self._N = 0
        self.types = []
        self.typeid = np.zeros(0, dtype=np.uint32)
        self.group = np.zeros((0, 2), dtype=np.uint32)

    @property
    def N(self):
        return self._N

    @N.setter
    def N(self, n):
        n = int(n)
        keep = min(n, self._N)
        typeid, group = np.zeros(n, dtype=np.uint32), np.zeros((n, 2), dtype=np.uint32)
        typeid[:keep], group[:keep] = self.typeid[:keep], self.group[:keep]
        self._N, self.typeid, self.group = n, typeid, group


class _Rank0:
    rank = 0
    num_ranks = 1


class Snapshot:
    """``snap.configuration.box`` + ``snap.particles.{N, types, position, velocity, typeid, mass, ...}``."""

    def __init__(self, communicator=None):
        self.communicator = communicator if communicator is not None else _Rank0()
        self.configuration = _Configuration()
        self.particles = _Particles()
        self.bonds = _Bonds()

    def _bonds_to_args(self):
        b = self.bonds
        group = np.ascontiguousarray(b.group, dtype=np.uint32)
        typeid = np.ascontiguousarray(b.typeid, dtype=np.uint32)
        if group.shape != (b.N, 2) or typeid.shape != (b.N,):
            raise ValueError(f"snapshot.bonds arrays do not match N={b.N}")
        if b.N and not b.types:
            raise RuntimeError("snapshot.bonds.types is empty")
        return group, typeid, list(b.types)

    def _to_dict(self):
        p = self.particles
        if not p.types:
            raise RuntimeError("snapshot.particles.types is empty")
        d = {"N": p.N, "types": list(p.types)}
        for name, (dt, k, _) in p._FIELDS.items():
            a = np.ascontiguousarray(getattr(p, name), dtype=dt)
            if a.shape != ((p.N, k) if k else (p.N,)):
                raise ValueError(f"snapshot.particles.{name} has shape {a.shape}, expected N={p.N}")
            d[name] = a
        return d

    @classmethod
    def _from_dict(cls, d, box, communicator=None):
        s = cls(communicator)
        s.configuration.box = box
        s.particles.N = int(d["N"])
        s.particles.types = list(d["types"])
        for name in s.particles._FIELDS:
            getattr(s.particles, name)[:] = d[name]
        return s

    def _set_bonds(self, b):
        self.bonds.N = int(b["N"])
        self.bonds.types = list(b["types"])
        self.bonds.typeid[:] = b["typeid"]
        self.bonds.group[:] = np.asarray(b["group"]).reshape(-1, 2)

    def wrap(self):
        """Wrap positions into the (orthorhombic) box and update images (hoomd/snapshot.py wrap)."""
        b = self.configuration._box
        L = b.L.copy()
        L[L == 0] = 1.0
        img = np.floor(self.particles.position / L + 0.5).astype(np.int32)
        if b.Lz == 0:
            img[:, 2] = 0
        self.particles.position -= img * L
        self.particles.image += img
