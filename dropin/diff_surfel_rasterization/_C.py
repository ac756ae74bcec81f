"""``diff_surfel_rasterization._C`` (ext.cpp:15-18): the compiled torch extension ``instascene_amd/_C_hip.so``
(csrc_torch/isr_torch_ext.cpp: the reference's three entry points on libinstascene_hip.so) when it is built, else the Python
mirror over the same C ABI (``instascene_amd.rasterizer``).  ``ISR_COMPILED_C=0`` forces the mirror."""
import os

COMPILED = False
if os.environ.get("ISR_COMPILED_C", "1") != "0":
    try:
        import torch  # noqa: F401  (the extension links torch's libraries: they must be loaded first)
        from instascene_amd._lib import lib as _lib
        _lib()                  # libinstascene_hip.so, found by the extension through its rpath, with every symbol checked
        from instascene_amd._C_hip import mark_visible, rasterize_gaussians, rasterize_gaussians_backward, set_mode  # noqa: F401
        COMPILED = True
    except Exception:       # not built, or built against another torch / library: the Python mirror serves the same names
        COMPILED = False
if not COMPILED:
    from instascene_amd.rasterizer import mark_visible, rasterize_gaussians, rasterize_gaussians_backward, set_mode  # noqa: F401
