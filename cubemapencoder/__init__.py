"""`import cubemapencoder` resolves to the MI355X implementation (materialrefgs_amd.cubemap_encoder -> libmrgs.so).

The reference imports `from cubemapencoder import CubemapEncoder` at the top of scene/gaussian_model.py:8; with this repository's
root on the path instead of the CUDA extension's package (submodules/cubemapencoder/cubemapencoder/__init__.py) that import binds
the HIP implementation unchanged.  See INTEGRATION.md.
"""
from materialrefgs_amd.cubemap_encoder import CubemapEncoder, MipCubemapEncoder, cubemap_encode  # noqa: F401

__all__ = ["CubemapEncoder", "MipCubemapEncoder", "cubemap_encode"]
