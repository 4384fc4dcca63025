"""model/ddpm_modules/joint_indi.py of the reference (inference half, HIP engine)."""
from ..samplers import InDISampler as IndiCustomT, InDISampler as IndiFullTranslation  # noqa: F401
from ..samplers import JointIndiSampler as JointIndi  # noqa: F401
