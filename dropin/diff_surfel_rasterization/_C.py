"""Mirror of the pybind module ``diff_surfel_rasterization._C`` (ext.cpp:15-18)."""
from instascene_amd.rasterizer import mark_visible, rasterize_gaussians, rasterize_gaussians_backward  # noqa: F401
