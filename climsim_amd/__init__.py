"""climsim_amd -- MI355X-native per-column physics-emulator path (hand-written HIP behind a C ABI).

Importing the package does not need a GPU; constructing an Emulator / wrapper does, and fails
loudly when the HIP library or device is missing (there is no CPU fallback in the product path).
"""
from . import _lib  # noqa: F401
from .emulator import Emulator  # noqa: F401
from .wrappers import NewModel_constraint, NewModel_constraint_ar, RNN_autoreg, model_wrapper  # noqa: F401

__all__ = ["Emulator", "NewModel_constraint", "NewModel_constraint_ar", "RNN_autoreg", "model_wrapper"]
