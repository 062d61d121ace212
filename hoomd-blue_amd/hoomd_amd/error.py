"""Exception types with the names the reference's Python layer uses (hoomd/error.py)."""


class DataAccessError(RuntimeError):
    """A loggable quantity was read before the object was attached to a simulation."""

    def __init__(self, data_name):
        self.data_name = data_name
        super().__init__(f"The property {data_name} is unavailable until the simulation runs for 0 or more steps.")


class IncompleteSpecificationError(ValueError):
    """A type parameter was left without a value (hoomd/data/typeparam.py semantics)."""


class MutabilityError(AttributeError):
    """An attribute that is fixed after attaching was assigned."""

    def __init__(self, attribute_name):
        super().__init__(f"The attribute {attribute_name} is immutable after operation has been added.")
