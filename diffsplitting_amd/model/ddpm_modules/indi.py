"""model/ddpm_modules/indi.py of the reference (inference half, HIP engine)."""
from ..samplers import InDISampler as InDI  # noqa: F401
