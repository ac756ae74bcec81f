"""Activation of the drop-in boundary: make the reference's import names resolve to this library
while its drivers (``train.py``, ``train_semantic.py``) run unmodified.

Reference import sites this serves:
  * ``from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer``
    (gaussian_renderer/__init__.py:14) and ``diff_surfel_rasterization._C`` (DSR/ext.cpp:15-18);
  * ``from simple_knn._C import distCUDA2`` (scene/gaussian_model.py:21);
  * ``from gaussian_renderer import render, network_gui`` (train.py:16), ``from gaussian_renderer import render``
    (train_semantic.py:7, spatial_track/modules/init_tracker.py:9);
  * ``from utils.contrastive_utils import *`` (train_semantic.py:9).

``python script.py`` puts the script's directory first on ``sys.path``, so the reference's own ``gaussian_renderer/`` and
``utils/`` always win over any shim directory - and they must stay importable (``network_gui``, ``utils.loss_utils``, the
visualisers ``contrastive_utils`` also defines).  So nothing is shadowed: :func:`install` adds ONE ``sys.meta_path``
finder that

  * serves the two native-extension packages (``diff_surfel_rasterization``, ``simple_knn`` and their ``_C``) from
    ``<repo>/dropin/`` ahead of any pip-installed CUDA build, and
  * lets the reference's own ``gaussian_renderer`` and ``utils.contrastive_utils`` modules be found and executed as
    usual, then rebinds ``render`` / ``contrastive_loss`` in them to the HIP implementations.

``install()`` also makes the driver's per-iteration ``torch.cuda.empty_cache()`` (train_semantic.py:208) act only under memory
pressure (:func:`empty_cache_under_pressure`; ``ISR_KEEP_EMPTY_CACHE=1`` opts out).

Activation, either of:
  * ``PYTHONPATH=<repo>/dropin:<repo> python train_semantic.py ...`` - ``dropin/sitecustomize.py`` calls ``install()``
    at interpreter start;
  * ``python -m instascene_amd.dropin train_semantic.py ...`` - installs, then runs the script as ``__main__``.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import importlib.util
import logging
import os
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SHIM_DIR = os.path.join(_REPO, "dropin")

# module name -> file under dropin/ (packages: their __init__.py)
_SHIMS = {
    "diff_surfel_rasterization": ("diff_surfel_rasterization/__init__.py", True),
    "diff_surfel_rasterization._C": ("diff_surfel_rasterization/_C.py", False),
    "simple_knn": ("simple_knn/__init__.py", True),
    "simple_knn._C": ("simple_knn/_C.py", False),
}


def _rebind_render(module):
    from .render import render
    module.render = render


def _rebind_contrastive(module):
    from .contrastive import contrastive_loss
    module.contrastive_loss = contrastive_loss


# module name -> patch applied right after the reference's own module body has run
_REBIND = {
    "gaussian_renderer": _rebind_render,
    "utils.contrastive_utils": _rebind_contrastive,
}


class _PatchingLoader(importlib.abc.Loader):
    """Runs the real loader of a reference module, then rebinds one name in it."""

    def __init__(self, inner, patch):
        self._inner, self._patch = inner, patch

    def create_module(self, spec):
        return self._inner.create_module(spec)

    def exec_module(self, module):
        self._inner.exec_module(module)
        self._patch(module)

    def __getattr__(self, name):            # get_source, is_package, get_filename ... of the real loader
        return getattr(self._inner, name)


class DropinFinder(importlib.abc.MetaPathFinder):
    def __init__(self):
        self._busy = False

    def find_spec(self, fullname, path=None, target=None):
        shim = _SHIMS.get(fullname)
        if shim is not None:
            rel, is_pkg = shim
            file = os.path.join(_SHIM_DIR, rel)
            return importlib.util.spec_from_file_location(
                fullname, file, submodule_search_locations=[os.path.dirname(file)] if is_pkg else None)
        patch = _REBIND.get(fullname)
        if patch is None or self._busy:
            return None
        self._busy = True                   # ask the finders behind this one (the reference checkout is on sys.path)
        try:
            spec = None
            for finder in sys.meta_path:
                if finder is self or not hasattr(finder, "find_spec"):
                    continue
                spec = finder.find_spec(fullname, path, target)
                if spec is not None:
                    break
        finally:
            self._busy = False
        if spec is None or spec.loader is None:
            return None
        spec.loader = _PatchingLoader(spec.loader, patch)
        return spec


_FINDER = None
_REAL_EMPTY_CACHE = None
EMPTY_CACHE = {"calls": 0, "honoured": 0}


def empty_cache_under_pressure(min_free_fraction: float = None):
    """Replace ``torch.cuda.empty_cache`` by a version that hands the cached memory back only when the device is actually short
    of it (free < ``min_free_fraction`` of the total, default 0.25, ``ISR_EMPTY_CACHE_MIN_FREE``).

    ``train_semantic.py:208`` empties the cache EVERY iteration - on the 24 GB cards the reference was written for that keeps
    fragmentation in check; on a 288 GB MI355X it makes every iteration return ~5 GB to the driver and ``hipMalloc`` it again
    (measured: 17 ms per iteration in a fresh process, 60-160 ms a few hundred iterations later, against 15 ms for the loop
    itself).  The library's own buffers are out of it either way (arena.py); this covers the reference's torch temporaries.
    Part of ``install()`` (``ISR_KEEP_EMPTY_CACHE=1`` leaves torch's function alone); idempotent; ``restore_empty_cache()`` undoes it."""
    global _REAL_EMPTY_CACHE
    import torch
    if _REAL_EMPTY_CACHE is not None:
        return
    frac = float(os.environ.get("ISR_EMPTY_CACHE_MIN_FREE", "0.25")) if min_free_fraction is None else float(min_free_fraction)
    real = _REAL_EMPTY_CACHE = torch.cuda.empty_cache
    torch.cuda._isr_real_empty_cache = real            # (arena.py's out-of-memory retry calls the original)

    def empty_cache():
        EMPTY_CACHE["calls"] += 1
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            free, total = torch.cuda.mem_get_info()
            if free >= frac * total:
                return
        EMPTY_CACHE["honoured"] += 1
        from . import arena
        arena.trim()                                   # the library's idle workspaces first, then torch's cache
        real()
    empty_cache.__doc__ = real.__doc__
    torch.cuda.empty_cache = empty_cache
    # a library that changes what a torch call does says so, once (review item 11)
    logging.getLogger("instascene_amd").warning(
        "instascene_amd.dropin: torch.cuda.empty_cache() now returns cached memory only when less than %.0f%% of the device is free "
        "(the reference's drivers call it every iteration; on an MI355X that costs more than the iteration). "
        "ISR_KEEP_EMPTY_CACHE=1 or dropin.restore_empty_cache() keeps torch's behaviour.", 100 * frac)


def restore_empty_cache():
    global _REAL_EMPTY_CACHE
    if _REAL_EMPTY_CACHE is not None:
        import torch
        torch.cuda.empty_cache = _REAL_EMPTY_CACHE
        if hasattr(torch.cuda, "_isr_real_empty_cache"):
            del torch.cuda._isr_real_empty_cache
        _REAL_EMPTY_CACHE = None


def install() -> DropinFinder:
    """Idempotent.  Also rebinds the names in modules that were imported before the call."""
    global _FINDER
    if os.environ.get("ISR_KEEP_EMPTY_CACHE", "0") != "1":
        empty_cache_under_pressure()
    if _REPO not in sys.path:
        sys.path.append(_REPO)              # `instascene_amd` itself
    if _FINDER is None:
        _FINDER = DropinFinder()
        sys.meta_path.insert(0, _FINDER)
    for name, patch in _REBIND.items():
        mod = sys.modules.get(name)
        if mod is not None:
            patch(mod)
    return _FINDER


def uninstall() -> None:
    global _FINDER
    restore_empty_cache()
    if _FINDER is not None and _FINDER in sys.meta_path:
        sys.meta_path.remove(_FINDER)
    _FINDER = None


def _main(argv):
    if not argv:
        print("usage: python -m instascene_amd.dropin <script.py> [args...]", file=sys.stderr)
        return 2
    import runpy
    install()
    script = os.path.abspath(argv[0])
    sys.argv = [script] + list(argv[1:])
    sys.path.insert(0, os.path.dirname(script))     # what `python script.py` does
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(_main(sys.argv[1:]))
