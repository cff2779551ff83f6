"""`import diff_surfel_rasterization` resolves to the MI355X rasterizer (materialrefgs_amd.rasterizer -> libmrgs.so).

The reference's render functions do `from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer`
(gaussian_renderer/__init__.py:15, gaussian_renderer/envgs_renderer.py, utils/mesh_utils.py); with this directory on the path
instead of the CUDA submodule's package (submodules/diff-surfel-rasterization/diff_surfel_rasterization/__init__.py) those
imports bind the HIP implementation unchanged.  See INTEGRATION.md section 1.
"""
from materialrefgs_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, _RasterizeGaussians,  # noqa: F401
                                          cpu_deep_copy_tuple, rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "cpu_deep_copy_tuple"]
