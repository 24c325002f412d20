from .checkpointer import CheckPointer  # noqa: F401
from .seed_torch import seed_torch  # noqa: F401
