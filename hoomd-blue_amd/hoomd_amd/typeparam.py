"""Per-type and per-type-pair parameters (role of hoomd/data/typeparam.py + parameterdicts.py).

``obj.params[('A', 'B')] = dict(epsilon=1.0, sigma=1.0)`` stores the value until the object is
attached, then forwards it to the C++ object; keys are unordered pairs of type names and may use
lists to set several pairs at once, as in the reference."""
import itertools

from .error import IncompleteSpecificationError


def _expand(key):
    if isinstance(key, str):
        raise KeyError(f"pair key must be a tuple of two type names, got {key!r}")
    a, b = key
    a = [a] if isinstance(a, str) else list(a)
    b = [b] if isinstance(b, str) else list(b)
    return [tuple(sorted(p)) for p in itertools.product(a, b)]


class PairTypeParameter:
    def __init__(self, name, convert, default=None):
        self.name = name
        self._convert = convert
        self._default = default
        self._values = {}
        self._setter = None
        self._getter = None
        self._types = None

    @property
    def default(self):
        return self._default

    @default.setter
    def default(self, value):
        self._default = self._convert(value)

    def __setitem__(self, key, value):
        value = self._convert(value)
        for pair in _expand(key):
            if self._types is not None:
                for t in pair:
                    if t not in self._types:
                        raise KeyError(f"Type {t} does not exist in the system")
            self._values[pair] = value
            if self._setter is not None:
                self._setter(pair, value)

    def __getitem__(self, key):
        pairs = _expand(key)
        out = {}
        for pair in pairs:
            if self._getter is not None:
                out[pair] = self._getter(pair)
            elif pair in self._values:
                out[pair] = self._values[pair]
            elif self._default is not None:
                out[pair] = self._default
            else:
                raise KeyError(pair)
        return out[pairs[0]] if len(pairs) == 1 else out

    def _attach(self, types, setter, getter):
        """Push every pair to the C++ object; a pair without a value is an error
        (hoomd/data/parameterdicts.py _validate_and_split / IncompleteSpecificationError)."""
        self._types = list(types)
        for pair in itertools.combinations_with_replacement(sorted(types), 2):
            if pair in self._values:
                v = self._values[pair]
            elif self._default is not None:
                v = self._default
            else:
                raise IncompleteSpecificationError(f"for {self.name} in pair {pair}: value is required")
            setter(pair, v)
        self._setter, self._getter = setter, getter

    def _detach(self):
        if self._getter is not None and self._types is not None:
            for pair in itertools.combinations_with_replacement(sorted(self._types), 2):
                self._values[pair] = self._getter(pair)
        self._setter = self._getter = self._types = None

    def to_base(self):
        return dict(self._values)
