"""Simulation box (hoomd/box.py): lengths Lx, Ly, Lz and tilt factors xy, xz, yz."""
import numpy as np


class Box:
    def __init__(self, Lx, Ly, Lz=0, xy=0, xz=0, yz=0):
        self.Lx, self.Ly, self.Lz, self.xy, self.xz, self.yz = (float(v) for v in (Lx, Ly, Lz, xy, xz, yz))

    @classmethod
    def cube(cls, L):
        return cls(L, L, L)

    @classmethod
    def square(cls, L):
        return cls(L, L, 0)

    @classmethod
    def from_box(cls, box):
        """Accepts a Box, a sequence [Lx, Ly, Lz(, xy, xz, yz)] or a dict, as hoomd.Box.from_box."""
        if isinstance(box, Box):
            return cls(box.Lx, box.Ly, box.Lz, box.xy, box.xz, box.yz)
        if isinstance(box, dict):
            return cls(**box)
        b = list(np.asarray(box, dtype=float).ravel())
        if len(b) not in (3, 6):
            raise ValueError("box must have 3 or 6 entries")
        return cls(*b)

    @property
    def dimensions(self):
        return 2 if self.Lz == 0 else 3

    @property
    def is2D(self):
        return self.Lz == 0

    @property
    def L(self):
        return np.array([self.Lx, self.Ly, self.Lz])

    @property
    def tilts(self):
        return np.array([self.xy, self.xz, self.yz])

    @property
    def volume(self):
        return self.Lx * self.Ly * (self.Lz if self.Lz else 1.0)

    def __eq__(self, other):
        return isinstance(other, Box) and np.array_equal(self.L, other.L) and np.array_equal(self.tilts, other.tilts)

    def __repr__(self):
        return f"hoomd.box.Box(Lx={self.Lx}, Ly={self.Ly}, Lz={self.Lz}, xy={self.xy}, xz={self.xz}, yz={self.yz})"
