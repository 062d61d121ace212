"""MI355X-native backend for HOOMD-blue's short-range MD step.

``hoomd_amd`` mirrors the slice of the ``hoomd`` Python API that drives the hot path
(``Simulation``, ``Snapshot``, ``Box``, ``device.GPU``, ``filter``, ``md.nlist.Cell``,
``md.pair.LJ``, ``md.Integrator``, ``md.methods.ConstantVolume``, ``tune.ParticleSorter``); the work
is done by C++ host classes (the ``_hoomd_amd`` extension module) that call hand-written gfx950
kernels through the C ABI declared in ``include/hoomd_amd.h``.  Importing the package requires both
shared libraries; there is no CPU fallback.
"""
__version__ = "0.1.0"

from . import _abi  # noqa: F401  (fails loudly when libhoomd_amd.so is missing)
from . import _hoomd_amd  # noqa: F401  (fails loudly when the host module is missing)
from . import device, error, filter, md, trigger, tune, write
from .box import Box
from .simulation import Simulation
from .snapshot import Snapshot

__all__ = ["Box", "Simulation", "Snapshot", "comm", "device", "error", "filter", "md", "trigger", "tune", "write"]


def __getattr__(name):
    # hoomd_amd.comm pulls in torch.distributed: loaded on first use only
    if name == "comm":
        import importlib
        return importlib.import_module(".comm", __name__)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
