"""Drop-in module for the reference's `from diff_gaussian_rasterization import
GaussianRasterizationSettings, GaussianRasterizer` (`gaussian_renderer/__init__.py:16`).
Backed by the MI355X HIP library through vcr_gaus_amd.rasterizer; nothing else lives here.

`set_num_dist(n)` / environment `VCR_NUM_DIST` stand for the fork's compile-time `NUM_DIST` (README.md:152-155): the number of
trailing output channels (0; 1 = depth distortion, read as `rendered_out[-1:]`; 2 = depth moments, read as `[-2:-1]`, `[-1:]`)."""
from vcr_gaus_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, get_num_dist,  # noqa: F401
                                     set_num_dist)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "set_num_dist", "get_num_dist"]
