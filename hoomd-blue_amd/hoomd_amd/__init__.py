"""MI355X-native backend for HOOMD-blue's short-range MD step.

``hoomd_amd`` mirrors the slice of the ``hoomd`` Python API that drives the hot path
(``Simulation``, ``Snapshot``, ``device.GPU``, ``md.nlist.Cell``, ``md.pair.LJ``, ``md.Integrator``,
``md.methods.ConstantVolume``); the work is done by C++ host classes (``_md`` extension module)
that call hand-written gfx950 kernels through the C ABI declared in ``include/hoomd_amd.h``.
"""
__version__ = "0.1.0"
