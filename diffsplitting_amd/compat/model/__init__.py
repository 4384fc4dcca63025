"""``import model`` -> diffsplitting_amd.model (reference module path; see diffsplitting_amd/_alias.py)."""
from diffsplitting_amd._alias import install

install(__name__)
