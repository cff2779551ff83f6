"""`import diff_surfel_tracing` resolves to the MI355X surfel tracer (materialrefgs_amd.surfel_tracing -> libmrgs.so).

The reference does `from diff_surfel_tracing import SurfelTracer, SurfelTracingSettings` (gaussian_renderer/optix_utils.py:7); that
package is an OptiX extension which is not part of the reference tree.  With this directory on the path the import binds the HIP
implementation; see INTEGRATION.md section 4g for what is and is not pinned about its arithmetic.
"""
from materialrefgs_amd.surfel_tracing import SurfelTracer, SurfelTracingSettings  # noqa: F401

__all__ = ["SurfelTracer", "SurfelTracingSettings"]
