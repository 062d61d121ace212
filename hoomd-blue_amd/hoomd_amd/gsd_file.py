"""GSD file layer: the container format of hoomd/extern/gsd.{h,c} (file version 2.0), restated.

Layout (all little endian): a 256-byte header (struct gsd_header, gsd.h:126-157), an index of 32-byte
entries (struct gsd_index_entry, gsd.h:165-187: frame, N, location, M, id, type, flags) sorted by frame
with unused entries zeroed, a name list of NUL-terminated chunk names (gsd.c:1497-1530), and the data
chunks.  A new file starts with room for 128 index entries and 1024 name bytes (gsd.c:1361-1425); when
either fills up it is written again, twice as large, at the end of the file and the header is updated
(the role of gsd_expand_file_index / gsd_append_names, gsd.c)."""
import os
import struct

import numpy as np

MAGIC = 0x65DF65DF65DF65DF
NAME_SIZE = 64
HEADER = struct.Struct("<QQQQQII64s64s80s")
ENTRY = np.dtype([("frame", "<u8"), ("N", "<u8"), ("location", "<i8"), ("M", "<u4"), ("id", "<u2"), ("type", "u1"), ("flags", "u1")])
assert HEADER.size == 256 and ENTRY.itemsize == 32
# enum gsd_type (gsd.h)
TYPES = {1: np.dtype("u1"), 2: np.dtype("<u2"), 3: np.dtype("<u4"), 4: np.dtype("<u8"), 5: np.dtype("i1"), 6: np.dtype("<i2"),
         7: np.dtype("<i4"), 8: np.dtype("<i8"), 9: np.dtype("<f4"), 10: np.dtype("<f8")}
TYPE_ID = {v: k for k, v in TYPES.items()}


def make_version(major, minor):
    """gsd_make_version (gsd.c): major << 16 | minor."""
    return (major << 16) | minor


class GSDFile:
    """``GSDFile(name, 'w', application=..., schema=..., schema_version=...)`` creates, ``'a'`` appends,
    ``'r'`` reads.  ``write_chunk`` + ``end_frame`` as gsd_write_chunk / gsd_end_frame; ``read_chunk(frame,
    name)`` as gsd_find_chunk + gsd_read_chunk; ``nframes`` as gsd_get_nframes."""

    def __init__(self, name, mode="r", application="", schema="", schema_version=(0, 0)):
        if mode not in ("r", "w", "a"):
            raise ValueError("mode must be 'r', 'w' or 'a'")
        self.name, self.mode = name, mode
        if mode == "w" or (mode == "a" and not os.path.exists(name)):
            self._f = open(name, "w+b")
            self._create(application, schema, make_version(*schema_version))
        else:
            self._f = open(name, "rb" if mode == "r" else "r+b")
        self._load()
        self._pending = []

    # ---- creation (gsd_initialize_file, gsd.c:1361-1425) ----
    def _create(self, application, schema, schema_version):
        index_entries, name_bytes = 128, 1024
        hdr = HEADER.pack(MAGIC, HEADER.size, index_entries, HEADER.size + ENTRY.itemsize * index_entries, name_bytes // NAME_SIZE,
                          schema_version, make_version(2, 0), application.encode()[:63], schema.encode()[:63], b"")
        self._f.write(hdr + bytes(ENTRY.itemsize * index_entries + name_bytes))
        self._f.flush()

    # ---- header, name list, index (gsd_initialize_handle, gsd.c:1440-1560) ----
    def _load(self):
        f = self._f
        f.seek(0)
        raw = f.read(HEADER.size)
        if len(raw) != HEADER.size:
            raise RuntimeError(f"{self.name}: not a GSD file")
        (magic, self._index_loc, self._index_alloc, self._names_loc, self._names_alloc, self.schema_version, self.gsd_version,
         app, schema, _) = HEADER.unpack(raw)
        if magic != MAGIC:
            raise RuntimeError(f"{self.name}: not a GSD file")
        if self.gsd_version >> 16 not in (1, 2):
            raise RuntimeError(f"{self.name}: unsupported GSD file version")
        self.application = app.split(b"\0")[0].decode()
        self.schema = schema.split(b"\0")[0].decode()
        f.seek(0, os.SEEK_END)
        size = f.tell()
        if self._names_loc + NAME_SIZE * self._names_alloc > size or self._index_loc + ENTRY.itemsize * self._index_alloc > size:
            raise RuntimeError(f"{self.name}: corrupt GSD file")
        f.seek(self._names_loc)
        blob = f.read(NAME_SIZE * self._names_alloc)
        if self.gsd_version < make_version(2, 0):
            names = [blob[i:i + NAME_SIZE].split(b"\0")[0] for i in range(0, len(blob), NAME_SIZE)]
            names = names[:names.index(b"")] if b"" in names else names
        else:
            names = blob.split(b"\0")
            names = names[:names.index(b"")]
        self._names = [n.decode() for n in names]
        self._name_bytes = sum(len(n) + 1 for n in names) if self.gsd_version >= make_version(2, 0) else NAME_SIZE * len(names)
        f.seek(self._index_loc)
        idx = np.frombuffer(f.read(ENTRY.itemsize * self._index_alloc), dtype=ENTRY)
        used = int(np.count_nonzero(idx["location"]))
        if used and (np.any(idx["location"][:used] == 0) or np.any(np.diff(idx["frame"][:used].astype(np.int64)) < 0)):
            raise RuntimeError(f"{self.name}: corrupt GSD index")
        self._index = idx[:used].copy()
        self.nframes = int(self._index["frame"][-1]) + 1 if used else 0

    # ---- reading ----
    def chunk_names(self, frame):
        e = self._index[self._index["frame"] == frame]
        return [self._names[i] for i in e["id"]]

    def find_chunk(self, frame, name):
        """The index entry or None (gsd_find_chunk)."""
        if name not in self._names:
            return None
        e = self._index[(self._index["frame"] == frame) & (self._index["id"] == self._names.index(name))]
        return e[-1] if len(e) else None

    def read_chunk(self, frame, name):
        e = self.find_chunk(frame, name)
        if e is None:
            raise KeyError(f"chunk {name} not found in frame {frame}")
        dt = TYPES[int(e["type"])]
        n, m = int(e["N"]), int(e["M"])
        self._f.seek(int(e["location"]))
        raw = self._f.read(n * m * dt.itemsize)
        if len(raw) != n * m * dt.itemsize:
            raise RuntimeError(f"{self.name}: chunk {name} runs past the end of the file")
        a = np.frombuffer(raw, dtype=dt)
        return a.reshape(n, m).copy() if m > 1 else a.copy()

    # ---- writing (gsd_write_chunk, gsd_end_frame) ----
    def write_chunk(self, name, data):
        if self.mode == "r":
            raise RuntimeError("file is open read-only")
        a = np.ascontiguousarray(data)
        if a.dtype not in TYPE_ID:
            a = a.astype(a.dtype.newbyteorder("<"))
        if a.ndim == 0:
            a = a.reshape(1)
        if a.ndim > 2 or a.size == 0:
            raise ValueError("chunks are non-empty one- or two-dimensional arrays")
        if len(name.encode()) >= NAME_SIZE * 2:
            raise ValueError("chunk name too long")
        self._pending.append((name, a))

    def end_frame(self):
        f = self._f
        new_names = []
        for name, _ in self._pending:
            if name not in self._names and name not in new_names:
                new_names.append(name)
        f.seek(0, os.SEEK_END)
        entries = np.zeros(len(self._pending), dtype=ENTRY)
        for k, (name, a) in enumerate(self._pending):
            entries[k] = (self.nframes, a.shape[0], f.tell(), a.shape[1] if a.ndim == 2 else 1,
                          (self._names + new_names).index(name), TYPE_ID[a.dtype], 0)
            f.write(a.tobytes())
        if not len(entries):  # an empty frame still counts: gsd.c does not allow it either
            raise RuntimeError("a frame needs at least one chunk")
        # the entries of a frame go into the index ordered by name id: gsd_find_chunk (hoomd/extern/gsd.c) bisects
        # the index on (frame, id), and gsd_end_frame sorts its buffered entries the same way before writing them
        entries = entries[np.argsort(entries["id"], kind="stable")]
        header_dirty = False
        if new_names:
            add = b"".join(n.encode() + b"\0" for n in new_names)
            if self._name_bytes + len(add) + 1 > NAME_SIZE * self._names_alloc:
                # move the name list to the end of the file, twice the room
                alloc = self._names_alloc
                while self._name_bytes + len(add) + 1 > NAME_SIZE * alloc:
                    alloc *= 2
                f.seek(self._names_loc)
                old = f.read(self._name_bytes)
                f.seek(0, os.SEEK_END)
                self._names_loc, self._names_alloc = f.tell(), alloc
                f.write(old + bytes(NAME_SIZE * alloc - len(old)))
                header_dirty = True
            f.seek(self._names_loc + self._name_bytes)
            f.write(add)
            self._name_bytes += len(add)
            self._names += new_names
        if len(self._index) + len(entries) > self._index_alloc:
            alloc = self._index_alloc
            while len(self._index) + len(entries) > alloc:
                alloc *= 2
            f.seek(0, os.SEEK_END)
            self._index_loc, self._index_alloc = f.tell(), alloc
            f.write(self._index.tobytes() + bytes(ENTRY.itemsize * (alloc - len(self._index))))
            header_dirty = True
        # data and names first, then the index entries that make them visible, then the header
        f.flush()
        f.seek(self._index_loc + ENTRY.itemsize * len(self._index))
        f.write(entries.tobytes())
        self._index = np.concatenate([self._index, entries])
        if header_dirty:
            f.seek(8)
            f.write(struct.pack("<QQQQ", self._index_loc, self._index_alloc, self._names_loc, self._names_alloc))
        f.flush()
        self._pending = []
        self.nframes += 1

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
