"""Host-side mirror of the reference's model surface for the hot path (model.MedPLIB / model.LISA)."""
from .config import MedPLIBConfig  # noqa: F401
