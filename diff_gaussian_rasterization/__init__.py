"""Drop-in module for the reference's `from diff_gaussian_rasterization import
GaussianRasterizationSettings, GaussianRasterizer` (`gaussian_renderer/__init__.py:16`).
Backed by the MI355X HIP library through vcr_gaus_amd.rasterizer; nothing else lives here."""
from vcr_gaus_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
