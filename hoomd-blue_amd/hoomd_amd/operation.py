"""The autotuner face of an operation (hoomd/operation.py:386-476, hoomd/Autotuned.h:26-89).

The reference sweeps block sizes (and threads per particle) per kernel at run time
(hoomd/Autotuner.h:169).  This backend is built for one architecture and launches every kernel with
fixed parameters, so there is nothing to sweep: `kernel_parameters` reports the values in use,
`is_tuning_complete` is true once the object is attached, and `tune_kernel_parameters` has nothing
left to do.  Parameters that the library does let a caller choose (threads per particle of the pair
force) can be set through `kernel_parameters` as in the reference."""
from .error import DataAccessError


class AutotunedObject:
    _cpp_obj = None

    def _kernel_parameters(self):
        """kernel name -> tuple of launch parameters in use"""
        return {}

    def _set_kernel_parameter(self, name, values):
        raise ValueError(f"kernel parameter {name!r} is fixed in this backend")

    @property
    def kernel_parameters(self):
        if self._cpp_obj is None:
            raise DataAccessError("kernel_parameters")
        return dict(self._kernel_parameters())

    @kernel_parameters.setter
    def kernel_parameters(self, parameters):
        if self._cpp_obj is None:
            raise DataAccessError("kernel_parameters")
        current = self._kernel_parameters()
        for name, values in dict(parameters).items():
            if name not in current:
                raise ValueError(f"no kernel named {name!r}; known: {sorted(current)}")
            values = tuple(int(v) for v in values)
            if values != tuple(current[name]):
                self._set_kernel_parameter(name, values)

    @property
    def is_tuning_complete(self):
        if self._cpp_obj is None:
            raise DataAccessError("is_tuning_complete")
        return True

    def tune_kernel_parameters(self):
        if self._cpp_obj is None:
            raise RuntimeError("Call Simulation.run() before tune_kernel_parameters.")
