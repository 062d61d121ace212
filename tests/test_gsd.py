"""GSD files: the container layer (hoomd_amd/gsd_file.py) against the reference's own file library
(hoomd/extern/gsd.c compiled into oracle/_ref/libgsd_ref.so by `make -C oracle ref`) in both directions,
against a file that library wrote (tests/golden/ref_written.gsd), and the "hoomd" schema round trip."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd")]
from hoomd_amd.gsd_file import ENTRY, TYPES, TYPE_ID, GSDFile, make_version  # noqa: E402

REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libgsd_ref.so")
GOLDEN = os.path.join(HERE, "golden", "ref_written.gsd")
READWRITE, READONLY, APPEND = 1, 2, 3  # enum gsd_open_flag (gsd.h)


class Ref:
    """ctypes front end of the reference library; the handle is an opaque block (struct gsd_handle is < 1 KiB)."""

    def __init__(self):
        self.lib = C.CDLL(REF_LIB)
        self.lib.gsd_find_chunk.restype = C.c_void_p
        self.lib.gsd_get_nframes.restype = C.c_uint64
        self.h = C.create_string_buffer(4096)

    def create(self, name, app, schema, version):
        assert self.lib.gsd_create_and_open(self.h, name.encode(), app.encode(), schema.encode(), C.c_uint32(version), APPEND, 0) == 0

    def open(self, name, flags):
        rc = self.lib.gsd_open(self.h, name.encode(), flags)
        assert rc == 0, f"gsd_open returned {rc}"

    def write(self, name, a):
        a = np.ascontiguousarray(a)
        n, m = (a.shape[0], a.shape[1]) if a.ndim == 2 else (a.shape[0], 1)
        assert self.lib.gsd_write_chunk(self.h, name.encode(), TYPE_ID[a.dtype], C.c_uint64(n), C.c_uint32(m), 0, a.ctypes.data_as(C.c_void_p)) == 0

    def end_frame(self):
        assert self.lib.gsd_end_frame(self.h) == 0

    def nframes(self):
        return int(self.lib.gsd_get_nframes(self.h))

    def read(self, frame, name):
        p = self.lib.gsd_find_chunk(self.h, C.c_uint64(frame), name.encode())
        if not p:
            return None
        e = np.frombuffer(C.string_at(p, ENTRY.itemsize), dtype=ENTRY)[0]
        dt = TYPES[int(e["type"])]
        out = np.empty(int(e["N"]) * int(e["M"]), dtype=dt)
        assert self.lib.gsd_read_chunk(self.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(p)) == 0
        return out.reshape(int(e["N"]), int(e["M"])) if int(e["M"]) > 1 else out

    def close(self):
        assert self.lib.gsd_close(self.h) == 0


needs_ref = pytest.mark.skipif(not os.path.exists(REF_LIB), reason="oracle/_ref/libgsd_ref.so not built (make -C oracle ref)")


def frames(n_frames, n_chunks, seed=0):
    """n_chunks differently named chunks of every type per frame: enough names and index entries to overflow the
    initial 1024 name bytes and 128 index entries."""
    rng = np.random.default_rng(seed)
    out = []
    for fr in range(n_frames):
        chunks = []
        for c in range(n_chunks):
            dt = TYPES[1 + (c + fr) % 10]
            shape = (int(rng.integers(1, 40)), int(rng.integers(1, 5)))
            a = (rng.random(shape) * 100).astype(dt)
            chunks.append((f"log/group_{c % 7}/a_rather_long_chunk_name_number_{c:03d}", a if shape[1] > 1 else a.reshape(-1)))
        out.append(chunks)
    return out


def same(a, b):
    return a is not None and a.dtype == b.dtype and np.array_equal(np.atleast_1d(a).reshape(b.shape), b)


@needs_ref
def test_files_written_here_are_read_by_the_reference_library(tmp_path):
    fn = str(tmp_path / "mine.gsd")
    data = frames(9, 40)
    with GSDFile(fn, "w", application="test", schema="hoomd", schema_version=(1, 4)) as f:
        for chunks in data:
            for name, a in chunks:
                f.write_chunk(name, a)
            f.end_frame()
    r = Ref()
    r.open(fn, READONLY)
    assert r.nframes() == len(data)
    for fr, chunks in enumerate(data):
        for name, a in chunks:
            assert same(r.read(fr, name), a), (fr, name)
    assert r.read(0, "no/such/chunk") is None
    r.close()


@needs_ref
def test_files_written_by_the_reference_library_are_read_here(tmp_path):
    fn = str(tmp_path / "ref.gsd")
    data = frames(9, 40, seed=4)
    r = Ref()
    r.create(fn, "ref app", "hoomd", make_version(1, 4))
    for chunks in data:
        for name, a in chunks:
            r.write(name, a)
        r.end_frame()
    r.close()
    with GSDFile(fn, "r") as f:
        assert (f.nframes, f.schema, f.application, f.schema_version) == (len(data), "hoomd", "ref app", make_version(1, 4))
        for fr, chunks in enumerate(data):
            assert sorted(f.chunk_names(fr)) == sorted(n for n, _ in chunks)
            for name, a in chunks:
                assert same(f.read_chunk(fr, name), a), (fr, name)
        with pytest.raises(KeyError):
            f.read_chunk(0, "no/such/chunk")


@needs_ref
def test_appending_to_each_others_files(tmp_path):
    fn = str(tmp_path / "both.gsd")
    a0, a1, a2 = np.arange(6, dtype=np.float32).reshape(2, 3), np.arange(5, dtype=np.uint32), np.array([7], dtype=np.uint64)
    with GSDFile(fn, "w", application="test", schema="hoomd", schema_version=(1, 4)) as f:
        f.write_chunk("particles/position", a0)
        f.end_frame()
    r = Ref()
    r.open(fn, APPEND)
    r.write("particles/typeid", a1)
    r.write("particles/position", a0 + 1)
    r.end_frame()
    r.close()
    with GSDFile(fn, "a") as f:
        assert f.nframes == 2
        f.write_chunk("configuration/step", a2)
        f.end_frame()
    r = Ref()
    r.open(fn, READONLY)
    assert r.nframes() == 3
    assert same(r.read(0, "particles/position"), a0) and same(r.read(1, "particles/position"), a0 + 1)
    assert same(r.read(1, "particles/typeid"), a1) and same(r.read(2, "configuration/step"), a2)
    assert r.read(2, "particles/position") is None
    r.close()


@needs_ref
def test_chunks_written_out_of_name_order_are_found_by_the_reference_library(tmp_path):
    """A name that first appears in a later frame gets the highest id; written before the older names, its index
    entry would precede theirs.  The reference looks chunks up by bisection on (frame, id), so the index must hold
    every frame's entries in id order whatever order they were written in."""
    fn = str(tmp_path / "order.gsd")
    rng = np.random.default_rng(8)
    old = [f"particles/chunk_{c:02d}" for c in range(12)]
    new = [f"log/new_{c:02d}" for c in range(5)]
    data = [[(n, rng.random(4).astype(np.float32)) for n in old]]
    # frame 1: new names first, then the old ones in reverse; frame 2: interleaved, one more new name in the middle
    data.append([(n, rng.random(3)) for n in new] + [(n, rng.random(4).astype(np.float32)) for n in reversed(old)])
    mixed = [n for pair in zip(reversed(new), old) for n in pair] + ["log/newest"] + old[len(new):]
    data.append([(n, rng.integers(0, 9, 6).astype(np.int32)) for n in mixed])
    with GSDFile(fn, "w", application="test", schema="hoomd", schema_version=(1, 4)) as f:
        for chunks in data:
            for name, a in chunks:
                f.write_chunk(name, a)
            f.end_frame()
        idx = f._index[: sum(len(c) for c in data)]
        assert all(np.all(np.diff(idx["id"][idx["frame"] == fr].astype(np.int64)) > 0) for fr in range(3))
    r = Ref()
    r.open(fn, READONLY)
    assert r.nframes() == 3
    for fr, chunks in enumerate(data):
        for name, a in chunks:
            assert same(r.read(fr, name), a), (fr, name)
    assert r.read(0, "log/new_00") is None and r.read(1, "log/newest") is None
    r.close()
    with GSDFile(fn, "r") as f:
        for fr, chunks in enumerate(data):
            for name, a in chunks:
                assert same(f.read_chunk(fr, name), a), (fr, name)


def test_golden_file_written_by_the_reference_library():
    # tests/golden/make_golden.py wrote it through hoomd/extern/gsd.c: two frames of a 5-particle system
    with GSDFile(GOLDEN, "r") as f:
        assert (f.nframes, f.schema, f.schema_version, f.gsd_version) == (2, "hoomd", make_version(1, 4), make_version(2, 0))
        assert f.read_chunk(0, "configuration/step")[0] == 100 and f.read_chunk(1, "configuration/step")[0] == 200
        assert np.array_equal(f.read_chunk(0, "particles/N"), np.array([5], dtype=np.uint32))
        pos = f.read_chunk(1, "particles/position")
        assert pos.dtype == np.float32 and pos.shape == (5, 3) and np.array_equal(pos, np.arange(15, dtype=np.float32).reshape(5, 3) * 0.25 + 1)
        assert f.find_chunk(1, "particles/typeid") is None and f.find_chunk(0, "particles/typeid") is not None
    from hoomd_amd.write.gsd import read_frame
    snap, step = read_frame(GOLDEN, -1)
    assert step == 200 and snap.particles.N == 5 and snap.particles.types == ["A", "Bee"]
    assert list(snap.particles.typeid) == [0, 1, 0, 1, 1]                      # taken from frame 0
    assert np.allclose(snap.particles.position, np.arange(15).reshape(5, 3) * 0.25 + 1)
    assert snap.configuration.box == [10.0, 12.0, 14.0, 0.0, 0.0, 0.0]
    assert np.all(snap.particles.mass == 1.0) and np.all(snap.particles.body == 0xFFFFFFFF)  # schema defaults
    assert snap.bonds.N == 2 and snap.bonds.types == ["polymer"] and snap.bonds.group.tolist() == [[0, 1], [1, 2]]


def test_rejects_what_is_not_a_gsd_file(tmp_path):
    fn = str(tmp_path / "junk.gsd")
    open(fn, "wb").write(b"\0" * 300)
    with pytest.raises(RuntimeError):
        GSDFile(fn, "r")
    open(fn, "wb").write(b"short")
    with pytest.raises(RuntimeError):
        GSDFile(fn, "r")
    with GSDFile(fn, "w", schema="hoomd") as f:
        with pytest.raises(RuntimeError):
            f.end_frame()
        with pytest.raises(ValueError):
            f.write_chunk("x", np.zeros((2, 2, 2)))


def test_hoomd_schema_round_trip(tmp_path):
    import hoomd_amd as hoomd
    from hoomd_amd.write.gsd import read_frame, write_frame
    rng = np.random.default_rng(2)
    s = hoomd.Snapshot()
    s.configuration.box = [8, 9, 10, 0.1, -0.2, 0.05]
    s.particles.N = 7
    s.particles.types = ["A", "B"]
    s.particles.position[:] = rng.uniform(-3, 3, (7, 3))
    s.particles.velocity[:] = rng.normal(size=(7, 3))
    s.particles.typeid[:] = [0, 1, 1, 0, 0, 1, 0]
    s.particles.image[:] = rng.integers(-2, 3, (7, 3))
    s.bonds.N = 3
    s.bonds.types = ["b0", "b1"]
    s.bonds.typeid[:] = [0, 1, 0]
    s.bonds.group[:] = [[0, 1], [1, 2], [5, 6]]
    fn = str(tmp_path / "traj.gsd")
    with GSDFile(fn, "w", application="t", schema="hoomd", schema_version=(1, 4)) as f:
        write_frame(f, s, 10, dynamic=("property",))
        names0 = f.chunk_names(0)
        s.particles.position[:] += 0.5
        s.particles.velocity[:] *= 2
        write_frame(f, s, 20, dynamic=("property",))
        names1 = f.chunk_names(1)
    # defaults (mass, charge, diameter, body) are left out; later frames hold only the dynamic group
    assert "particles/mass" not in names0 and "particles/body" not in names0 and "particles/velocity" in names0
    assert sorted(names1) == ["configuration/box", "configuration/step", "particles/N", "particles/position"]
    a, step = read_frame(fn, 1)
    assert step == 20 and a.particles.types == ["A", "B"]
    assert np.array_equal(a.particles.position, s.particles.position.astype(np.float32).astype(np.float64))
    assert np.array_equal(a.particles.velocity, (s.particles.velocity / 2).astype(np.float32).astype(np.float64))  # frame 0's
    assert np.array_equal(a.particles.typeid, s.particles.typeid) and np.array_equal(a.particles.image, s.particles.image)
    assert np.allclose(a.configuration.box, [8, 9, 10, 0.1, -0.2, 0.05], rtol=1e-7)
    assert a.bonds.N == 3 and a.bonds.types == ["b0", "b1"] and np.array_equal(a.bonds.group, s.bonds.group)
    with pytest.raises(IndexError):
        read_frame(fn, 2)
