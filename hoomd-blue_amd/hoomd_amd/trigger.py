"""Triggers (hoomd/trigger.py, hoomd/Trigger.h): callables of the time step."""


class Trigger:
    def compute(self, timestep):
        raise NotImplementedError

    def __call__(self, timestep):
        return bool(self.compute(int(timestep)))


class Periodic(Trigger):
    """True every ``period`` steps: (timestep - phase) % period == 0 (hoomd/Trigger.h:64-100)."""

    def __init__(self, period, phase=0):
        if int(period) < 1:
            raise ValueError("period must be positive")
        self.period, self.phase = int(period), int(phase)

    def compute(self, timestep):
        return (timestep - self.phase) % self.period == 0

    def __eq__(self, other):
        return isinstance(other, Periodic) and (self.period, self.phase) == (other.period, other.phase)

    def __hash__(self):
        return hash(("Periodic", self.period, self.phase))


class On(Trigger):
    """True at one time step (hoomd/Trigger.h OnTrigger)."""

    def __init__(self, timestep):
        self.timestep = int(timestep)

    def compute(self, timestep):
        return timestep == self.timestep


def as_trigger(value):
    """An int means Periodic(value), as in the reference's trigger conversion (hoomd/trigger.py)."""
    if isinstance(value, Trigger):
        return value
    if isinstance(value, int):
        return Periodic(value)
    raise TypeError("expected a Trigger or an integer period")
