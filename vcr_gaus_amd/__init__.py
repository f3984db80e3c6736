"""MI355X-native differentiable Gaussian-splatting rasterizer + D-Normal training hot path.

Host side mirrors the reference's operator surface (`gaussian_renderer.render`,
`GaussianModel`, `diff_gaussian_rasterization.GaussianRasterizer`); all arithmetic on the hot
path runs in hand-written HIP kernels (csrc/) behind the C ABI declared in include/vcr_raster.h.
"""
__version__ = "0.2.0"
