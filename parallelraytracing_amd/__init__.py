"""parallelraytracing_amd — MI355X-native wavefront path tracer behind the reference's Renderer interface.

Only the hot path (ray generation -> BVH closest hit -> shade/scatter -> film accumulate) lives here, as
hand-written HIP kernels in csrc/ behind the C-ABI of include/prt.h.  Importing the package loads
csrc/libprt.so and fails loudly when it has not been built: there is no CPU or Python fallback.
"""
from . import capi

capi.lib()  # fail loudly if the HIP extension is missing

from .renderer import (Camera, Film, HipWavefrontGroupRenderer, HipWavefrontRenderer, Mesh, PrtError, Scene,  # noqa: E402
                       glm_normalize, make_transform, write_pfm, write_ppm)
from . import dist, scenes  # noqa: E402

__all__ = ["Camera", "Film", "HipWavefrontGroupRenderer", "HipWavefrontRenderer", "Mesh", "PrtError", "Scene", "capi", "dist", "glm_normalize",
           "make_transform", "scenes", "write_pfm", "write_ppm"]
