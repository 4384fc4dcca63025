"""MI355X-native sampling engine behind the DiffSplitting entry points.

``diffsplitting_amd.model.networks.define_G`` / ``model.create_model`` /
``DDPM.test()`` / ``split.py`` mirror the reference's Python API; the hot path
(UNet forward, reverse-sampling loop, tile gather/stitch) runs in
``libdsx.so`` (hand-written gfx950 HIP, C ABI in ``include/dsx.h``).
"""
__version__ = "0.1.0"
