"""model/sr3_modules/diffusion.py of the reference (sampling half, HIP engine)."""
from ..samplers import GaussianSampler as GaussianDiffusion  # noqa: F401
from ...engine import make_beta_schedule  # noqa: F401
