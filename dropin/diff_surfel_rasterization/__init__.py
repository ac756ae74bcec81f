"""Drop-in for the reference's ``diff_surfel_rasterization`` package: put ``<repo>/dropin`` (and ``<repo>``)
ahead of the reference checkout on PYTHONPATH and ``gaussian_renderer/__init__.py:14`` imports this."""
from instascene_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                       rasterize_gaussians_autograd as rasterize_gaussians)
from . import _C  # noqa: F401
