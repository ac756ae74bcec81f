"""``diff_surfel_rasterization._C`` (ext.cpp:15-18): the compiled torch extension ``instascene_amd/_C_hip.so``
(csrc_torch/isr_torch_ext.cpp: the reference's three entry points on libinstascene_hip.so) when it is built, else the Python
mirror over the same C ABI (``instascene_amd.rasterizer``).  ``ISR_COMPILED_C=0`` forces the mirror.

One source of truth for the arithmetic mode / tile lists: ``instascene_amd.rasterizer`` (``set_mode`` / ``get_mode``).  When the
extension loads it is handed that mode, and ``_C.set_mode`` here is a wrapper of ``rasterizer.set_mode`` (which forwards to the
extension) - a process never runs its ``GaussianRasterizer`` forwards and its direct ``_C`` calls in different modes."""
import logging
import os

from instascene_amd import rasterizer as _rz

COMPILED = False
FALLBACK_REASON = None          # why the mirror serves the names although the extension was wanted (None: it does not, or by choice)
if os.environ.get("ISR_COMPILED_C", "1") != "0":
    try:
        import torch  # noqa: F401  (the extension links torch's libraries: they must be loaded first)
        from instascene_amd._lib import lib as _lib
        _lib()                  # libinstascene_hip.so, found by the extension through its rpath, with every symbol checked
        from instascene_amd import _C_hip as _ext
        from instascene_amd._C_hip import mark_visible, rasterize_gaussians, rasterize_gaussians_backward  # noqa: F401
        _ext.set_mode(_rz.get_mode())          # an extension imported after rasterizer.set_mode() starts in that mode
        COMPILED = True
    except Exception as _e:     # not built, or built against another torch / library: the Python mirror serves the same names
        COMPILED = False
        FALLBACK_REASON = repr(_e)
        # said once (this module is imported once): the two bindings are bit-identical, but a maintainer should know which one runs
        logging.getLogger("instascene_amd").warning(
            "diff_surfel_rasterization._C: the compiled extension instascene_amd/_C_hip.so did not load (%s); the Python binding over "
            "the same C ABI serves rasterize_gaussians / rasterize_gaussians_backward / mark_visible instead "
            "(rebuild: python -m instascene_amd.csrc_torch.build --force)", FALLBACK_REASON)
if not COMPILED:
    from instascene_amd.rasterizer import mark_visible, rasterize_gaussians, rasterize_gaussians_backward  # noqa: F401


def set_mode(mode: str):
    """"exact" | "fast_reflists" | "fast": sets the mode of BOTH bindings (instascene_amd.rasterizer.set_mode)."""
    _rz.set_mode(mode)
