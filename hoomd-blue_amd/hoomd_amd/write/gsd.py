"""GSD trajectory writer and reader for the "hoomd" schema, version 1.4 (hoomd/write/gsd.py,
hoomd/GSDDumpWriter.cc, hoomd/GSDReader.cc).  Floating-point chunks are 32-bit, as the schema says."""
import numpy as np

from .. import filter as _filter
from ..gsd_file import GSDFile
from ..snapshot import Snapshot
from ..trigger import as_trigger

_SCHEMA_VERSION = (1, 4)
# chunk -> (dtype, columns, default), hoomd/GSDDumpWriter.cc:513-760 and the schema's defaults
_PARTICLE = {
    "typeid": (np.uint32, 1, 0), "mass": (np.float32, 1, 1.0), "charge": (np.float32, 1, 0.0), "diameter": (np.float32, 1, 1.0),
    "body": (np.int32, 1, -1), "position": (np.float32, 3, 0.0), "velocity": (np.float32, 3, 0.0), "image": (np.int32, 3, 0),
}
_GROUPS = {"attribute": ("typeid", "mass", "charge", "diameter", "body"), "property": ("position",), "momentum": ("velocity", "image")}


def _type_table(names):
    """NT x (longest + 1) bytes, NUL padded (GSDDumpWriter::writeTypeMapping, hoomd/GSDDumpWriter.cc:426-449)."""
    width = max(len(n.encode()) for n in names) + 1
    t = np.zeros((len(names), width), dtype=np.uint8)
    for i, n in enumerate(names):
        b = n.encode()
        t[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
    return t


def _type_names(table):
    table = np.atleast_2d(table)
    return [bytes(row).split(b"\0")[0].decode() for row in table.astype(np.uint8)]


def write_frame(f, snapshot, timestep, dynamic=("property",), tags=None):
    """One frame.  The first frame of a file holds everything that differs from the schema's defaults; later
    frames hold the step, the box, N and the dynamic groups (hoomd/GSDDumpWriter.cc:454-508, :513-900).
    ``tags``: write only these particles (a filter's tags, ascending); bonds are then left out unless all are kept."""
    first = f.nframes == 0
    p = snapshot.particles
    sel = slice(None) if tags is None else np.asarray(tags, dtype=np.int64)
    n = p.N if tags is None else len(sel)
    box = snapshot.configuration._box
    f.write_chunk("configuration/step", np.array([timestep], dtype=np.uint64))
    if first:
        f.write_chunk("configuration/dimensions", np.array([box.dimensions], dtype=np.uint8))
    f.write_chunk("configuration/box", np.array([box.Lx, box.Ly, box.Lz, box.xy, box.xz, box.yz], dtype=np.float32))
    f.write_chunk("particles/N", np.array([n], dtype=np.uint32))
    if n == 0:
        f.end_frame()
        return
    if first or "attribute" in dynamic:
        f.write_chunk("particles/types", _type_table(list(p.types)))
    for group, names in _GROUPS.items():
        for name in names:
            dt, cols, default = _PARTICLE[name]
            a = getattr(p, name)[sel]
            if name == "body":
                a = a.astype(np.int64)
                a[a == 0xFFFFFFFF] = -1
            a = a.astype(dt)
            nondefault = bool(np.any(a != dt(default)))
            if (first and nondefault) or (not first and group in dynamic and (nondefault or f.find_chunk(0, "particles/" + name) is not None)):
                f.write_chunk("particles/" + name, a)
    b = snapshot.bonds
    if (first or "topology" in dynamic) and tags is None:
        f.write_chunk("bonds/N", np.array([b.N], dtype=np.uint32))
        if b.types:
            f.write_chunk("bonds/types", _type_table(list(b.types)))
        if b.N:
            f.write_chunk("bonds/typeid", b.typeid.astype(np.uint32))
            f.write_chunk("bonds/group", b.group.astype(np.uint32))
    f.end_frame()


def read_frame(filename, frame=-1):
    """(Snapshot, time step) of one frame; a chunk missing from the frame is taken from frame 0, then from the
    schema's default (hoomd/GSDReader.cc:96-160 readChunk)."""
    with GSDFile(filename, "r") as f:
        if f.schema != "hoomd":
            raise RuntimeError(f"{filename}: schema is {f.schema!r}, not 'hoomd'")
        if f.schema_version >> 16 != 1:
            raise RuntimeError(f"{filename}: unsupported hoomd schema version")
        if f.nframes == 0:
            raise RuntimeError(f"{filename}: no frames")
        if frame < 0:
            frame += f.nframes
        if not 0 <= frame < f.nframes:
            raise IndexError(f"frame {frame} out of range (file has {f.nframes})")

        def chunk(name, need_rows=None):
            for fr in (frame, 0):
                if f.find_chunk(fr, name) is not None:
                    a = f.read_chunk(fr, name)
                    if need_rows is None or a.shape[0] == need_rows:
                        return a
            return None

        snap = Snapshot()
        step = chunk("configuration/step")
        box = chunk("configuration/box")
        if box is None:
            raise RuntimeError(f"{filename}: frame {frame} has no box")
        dims = chunk("configuration/dimensions")
        box = [float(v) for v in box]
        if dims is not None and int(dims[0]) == 2:
            box[2] = 0.0
        snap.configuration.box = box
        n = chunk("particles/N")
        n = int(n[0]) if n is not None else 0
        snap.particles.N = n
        types = chunk("particles/types")
        snap.particles.types = _type_names(types) if types is not None else ["A"]
        for name, (dt, cols, default) in _PARTICLE.items():
            a = chunk("particles/" + name, n)
            if a is None:
                continue
            if name == "body":
                a = a.astype(np.int64)
                a[a < 0] = 0xFFFFFFFF
            getattr(snap.particles, name)[:] = a
        nb = chunk("bonds/N")
        nb = int(nb[0]) if nb is not None else 0
        snap.bonds.N = nb
        bt = chunk("bonds/types")
        snap.bonds.types = _type_names(bt) if bt is not None else []
        if nb:
            tid, grp = chunk("bonds/typeid", nb), chunk("bonds/group", nb)
            if tid is not None:
                snap.bonds.typeid[:] = tid
            if grp is None:
                raise RuntimeError(f"{filename}: bonds/N = {nb} but no bonds/group")
            snap.bonds.group[:] = grp
        return snap, (int(step[0]) if step is not None else 0)


class GSD:
    """``hoomd.write.GSD(trigger, filename, filter=All(), mode='ab', truncate=False, dynamic=None)``
    (hoomd/write/gsd.py:24-260).  ``GSD.write(state, filename, filter, mode)`` writes one frame."""

    def __init__(self, trigger, filename, filter=None, mode="ab", truncate=False, dynamic=None):
        if mode not in ("wb", "xb", "ab"):
            raise ValueError("mode must be 'wb', 'xb' or 'ab'")
        self.trigger = as_trigger(trigger)
        self.filename = str(filename)
        self.filter = filter if filter is not None else _filter.All()
        self.mode, self.truncate = mode, bool(truncate)
        dynamic = ["property"] if dynamic is None else list(dynamic)
        for d in dynamic:
            if d not in ("attribute", "property", "momentum", "topology"):
                raise ValueError(f"unknown dynamic group {d}")
        self.dynamic = tuple(dynamic)
        self._file = None

    def _open(self):
        import os
        if self.mode == "xb" and os.path.exists(self.filename):
            raise FileExistsError(self.filename)
        m = "a" if self.mode == "ab" and not self.truncate else "w"
        self._file = GSDFile(self.filename, m, application="hoomd_amd (HOOMD-blue 4.1.0 API)", schema="hoomd", schema_version=_SCHEMA_VERSION)
        if self._file.schema != "hoomd":
            raise RuntimeError(f"{self.filename}: existing file has another schema")

    def _write(self, sim):
        if self._file is None:
            self._open()
        elif self.truncate:
            self._file.close()
            self._open()
        state = sim.state
        tags = self.filter(state)
        write_frame(self._file, state.get_snapshot(), sim.timestep, self.dynamic, None if tags is None else np.sort(np.asarray(tags)))

    def flush(self):
        if self._file is not None:
            self._file._f.flush()

    def _detach(self):
        if self._file is not None:
            self._file.close()
            self._file = None

    @staticmethod
    def write(state, filename, filter=None, mode="wb"):
        """One frame with everything in it (hoomd/write/gsd.py:213-245)."""
        w = GSD(1, filename, filter=filter, mode=mode, dynamic=["attribute", "property", "momentum", "topology"])
        w._write(state._sim)
        w._detach()
