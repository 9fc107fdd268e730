"""Drop-in for the reference's `diff_gaussian_rasterization` import
(gaussian_renderer/__init__.py:15).  Put this repo's root on sys.path and the reference's
gaussian_renderer/__init__.py runs unmodified on MI355X.  Implementation:
event_3dgs_amd/rasterizer.py over the C ABI in include/e3dgs_hip.h."""
from event_3dgs_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                        rasterize_gaussians)
