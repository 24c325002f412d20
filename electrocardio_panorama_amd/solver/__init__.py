from .solver import Solver  # noqa: F401
