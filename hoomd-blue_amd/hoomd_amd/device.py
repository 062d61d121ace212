"""Execution devices (hoomd/device.py).  This backend drives one MI355X per process."""
import os

from . import _hoomd_amd as _cpp


class Device:
    def __init__(self, communicator=None):
        self.communicator = communicator


class GPU(Device):
    """Select a GPU (hoomd/device.py:300-420).  ``gpu_id=None`` picks LOCAL_RANK when launched by
    ``torch.distributed.run``, else device 0."""

    def __init__(self, gpu_id=None, communicator=None):
        super().__init__(communicator)
        n = _cpp.device_count()
        if n == 0:
            raise RuntimeError("no HIP device is visible: hoomd_amd.device.GPU needs a MI355X")
        if gpu_id is None:
            gpu_id = int(os.environ.get("LOCAL_RANK", "0")) % n
        self.gpu_id = gpu_id
        _cpp.set_device(gpu_id)
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.set_device(gpu_id)
        except ImportError:
            pass

    @property
    def devices(self):
        return [_cpp.device_name(self.gpu_id)]

    @staticmethod
    def is_available():
        return _cpp.device_count() > 0

    @staticmethod
    def get_available_devices():
        return [_cpp.device_name(i) for i in range(_cpp.device_count())]


class CPU(Device):
    """The reference's CPU device has no counterpart here: this package IS the GPU backend."""

    def __init__(self, *a, **k):
        raise NotImplementedError("hoomd_amd implements hoomd.device.GPU only; use the reference for hoomd.device.CPU")


def auto_select(communicator=None):
    return GPU(communicator=communicator)
