"""Particle filters (hoomd/filter/*.py): select the tags an operation acts on."""
import numpy as np


class ParticleFilter:
    def __call__(self, state):
        raise NotImplementedError

    def __hash__(self):
        return hash(type(self).__name__)

    def __eq__(self, other):
        return type(self) is type(other)


class All(ParticleFilter):
    def __call__(self, state):
        return None  # None = every particle (identity group on the device)


class Null(ParticleFilter):
    def __call__(self, state):
        return np.zeros(0, dtype=np.uint32)


class Tags(ParticleFilter):
    def __init__(self, tags):
        self.tags = np.unique(np.asarray(tags, dtype=np.uint32))

    def __call__(self, state):
        return self.tags

    def __hash__(self):
        return hash(self.tags.tobytes())

    def __eq__(self, other):
        return type(self) is type(other) and np.array_equal(self.tags, other.tags)


class Type(ParticleFilter):
    def __init__(self, types):
        self.types = frozenset(types)

    def __call__(self, state):
        snap = state.get_snapshot()
        ids = [i for i, t in enumerate(snap.particles.types) if t in self.types]
        return np.nonzero(np.isin(snap.particles.typeid, ids))[0].astype(np.uint32)

    def __hash__(self):
        return hash(self.types)

    def __eq__(self, other):
        return type(self) is type(other) and self.types == other.types
