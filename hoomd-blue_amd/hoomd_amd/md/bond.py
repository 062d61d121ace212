"""Bond forces (hoomd/md/bond.py): the ``Bond`` base and ``FENEWCA``."""
from .. import _hoomd_amd as _cpp
from ..error import IncompleteSpecificationError
from .pair import Force


class _BondTypeParameter:
    """``params['A-A'] = dict(...)`` keyed by bond type name (hoomd/data/typeparam.py TypeParameter)."""

    def __init__(self, name, convert):
        self.name = name
        self._convert = convert
        self._values = {}
        self._types = None
        self._setter = self._getter = None

    def __setitem__(self, key, value):
        value = self._convert(value)
        keys = [key] if isinstance(key, str) else list(key)
        for k in keys:
            if self._types is not None and k not in self._types:
                raise KeyError(f"Type {k} does not exist in the system")
            self._values[k] = value
            if self._setter is not None:
                self._setter(k, value)

    def __getitem__(self, key):
        if self._getter is not None:
            return self._getter(key)
        return self._values[key]

    def _attach(self, types, setter, getter):
        self._types = list(types)
        for t in types:
            if t not in self._values:
                raise IncompleteSpecificationError(f"for {self.name} in bond type {t}: value is required")
            setter(t, self._values[t])
        self._setter, self._getter = setter, getter

    def _detach(self):
        if self._getter is not None:
            for t in self._types:
                self._values[t] = self._getter(t)
        self._types = self._setter = self._getter = None


class Bond(Force):
    """Base of the bond forces (hoomd/md/bond.py:24-50)."""

    _cpp_class_name = None

    def __init__(self):
        self.params = _BondTypeParameter("params", self._convert_params)

    def _convert_params(self, value):
        raise NotImplementedError

    def _attach(self, sim):
        self._sim = sim
        self._cpp_obj = getattr(_cpp, self._cpp_class_name + "GPU")(sim._sysdef)
        types = sim._state.bond_types
        tid = {t: i for i, t in enumerate(types)}
        self.params._attach(types, lambda t, v: self._cpp_obj.setParams(tid[t], v), lambda t: self._cpp_obj.getParams(tid[t]))

    def _detach(self):
        self.params._detach()
        self._cpp_obj = None
        self._sim = None


class FENEWCA(Bond):
    r"""FENE springs with a WCA core (hoomd/md/bond.py:107-175): ``params[bond type] = dict(k, r0, epsilon,
    sigma, delta)``; V = -k r0^2/2 ln(1 - ((r - delta)/r0)^2) + WCA(r - delta)."""

    _cpp_class_name = "PotentialBondFENE"

    def _convert_params(self, value):
        missing = {"k", "r0", "epsilon", "sigma", "delta"} - set(value)
        if missing:
            raise ValueError(f"FENEWCA params need {sorted(missing)}")
        return {k: float(value[k]) for k in ("k", "r0", "epsilon", "sigma", "delta")}
