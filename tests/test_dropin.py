"""The drop-in boundary in a real launch (reference import sites train.py:15-16, train_semantic.py:7-10,
spatial_track/modules/init_tracker.py:9): a skeleton checkout - written here, holding none of the reference's text - is
run with ``python train_like.py`` under the documented environment, and the reference's own packages must stay whole
while ``render`` / ``contrastive_loss`` / the two native extensions resolve to this library."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _skeleton(tmp_path):
    """A directory shaped like the reference checkout: its own gaussian_renderer/ package (render + network_gui) and a
    namespace package utils/ (the reference's utils/ has no __init__.py) with contrastive_utils + loss_utils."""
    (tmp_path / "gaussian_renderer").mkdir()
    (tmp_path / "utils").mkdir()
    (tmp_path / "scene").mkdir()
    (tmp_path / "gaussian_renderer" / "__init__.py").write_text(textwrap.dedent("""
        from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        from utils.point_helpers import helper
        STUB = "skeleton gaussian_renderer"
        def render(*a, **k):
            raise AssertionError("the skeleton's own render() must have been replaced")
    """))
    (tmp_path / "gaussian_renderer" / "network_gui.py").write_text("conn = None\ndef try_connect():\n    return 'gui'\n")
    (tmp_path / "utils" / "point_helpers.py").write_text("def helper():\n    return 'own utils module'\n")
    (tmp_path / "utils" / "loss_utils.py").write_text("def l1_loss(a, b):\n    return 'own l1'\n")
    (tmp_path / "utils" / "contrastive_utils.py").write_text(textwrap.dedent("""
        def contrastive_loss(*a, **k):
            raise AssertionError("the skeleton's own contrastive_loss must have been replaced")
        def feature_to_rgb(x):
            return 'own visualiser'
    """))
    (tmp_path / "scene" / "__init__.py").write_text("from simple_knn._C import distCUDA2\n")
    (tmp_path / "train_like.py").write_text(textwrap.dedent("""
        import json, sys
        from utils.loss_utils import l1_loss
        from gaussian_renderer import render, network_gui
        from utils.contrastive_utils import *
        from scene import distCUDA2
        import gaussian_renderer, diff_surfel_rasterization, simple_knn._C
        import instascene_amd.render, instascene_amd.contrastive, instascene_amd.rasterizer, instascene_amd.knn
        print(json.dumps({
            "render": render is instascene_amd.render.render,
            "render_attr": gaussian_renderer.render is instascene_amd.render.render,
            "loss": contrastive_loss is instascene_amd.contrastive.contrastive_loss,
            "visualiser": feature_to_rgb(0),
            "gui": network_gui.try_connect(),
            "own_pkg": gaussian_renderer.STUB,
            "l1": l1_loss(0, 0),
            "rasterizer": diff_surfel_rasterization.GaussianRasterizer is instascene_amd.rasterizer.GaussianRasterizer,
            # the compiled extension (instascene_amd/_C_hip.so) when it is built, else the Python mirror over the same C ABI
            "c_ext": (diff_surfel_rasterization._C.rasterize_gaussians is sys.modules["instascene_amd._C_hip"].rasterize_gaussians)
                     if diff_surfel_rasterization._C.COMPILED else
                     (diff_surfel_rasterization._C.rasterize_gaussians is instascene_amd.rasterizer.rasterize_gaussians),
            "knn": distCUDA2 is instascene_amd.knn.distCUDA2,
            "argv": sys.argv[1:],
        }))
    """))
    return tmp_path


def _check(out):
    rec = json.loads(out.strip().splitlines()[-1])
    assert rec == {"render": True, "render_attr": True, "loss": True, "visualiser": "own visualiser", "gui": "gui",
                   "own_pkg": "skeleton gaussian_renderer", "l1": "own l1", "rasterizer": True, "c_ext": True, "knn": True,
                   "argv": ["--flag", "7"]}, rec


def _env(extra_path):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = extra_path
    return env


def test_documented_pythonpath_activation(tmp_path):
    """PYTHONPATH=<repo>/dropin, then the unmodified driver."""
    tree = _skeleton(tmp_path)
    r = subprocess.run([sys.executable, "train_like.py", "--flag", "7"], cwd=tree, env=_env(os.path.join(ROOT, "dropin")),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    _check(r.stdout)


def test_module_runner_activation(tmp_path):
    """python -m instascene_amd.dropin train_like.py ...  (no sitecustomize involved)."""
    tree = _skeleton(tmp_path)
    env = _env(ROOT)
    env["ISR_DROPIN"] = "0"
    r = subprocess.run([sys.executable, "-m", "instascene_amd.dropin", "train_like.py", "--flag", "7"], cwd=tree, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    _check(r.stdout)


def test_without_activation_the_skeleton_is_untouched(tmp_path):
    tree = _skeleton(tmp_path)
    env = _env(os.path.join(ROOT, "dropin"))
    env["ISR_DROPIN"] = "0"
    code = "import utils.contrastive_utils as u, sys; sys.exit(0 if u.contrastive_loss.__module__ == 'utils.contrastive_utils' else 1)"
    r = subprocess.run([sys.executable, "-c", code], cwd=tree, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr


def test_install_rebinds_already_imported_modules(tmp_path, monkeypatch):
    import importlib
    import types
    from instascene_amd import dropin
    fake = types.ModuleType("gaussian_renderer")
    fake.render = lambda *a: None
    monkeypatch.setitem(sys.modules, "gaussian_renderer", fake)
    try:
        dropin.install()
        import instascene_amd.render
        assert fake.render is instascene_amd.render.render
        assert importlib.import_module("diff_surfel_rasterization").GaussianRasterizer is not None
    finally:
        dropin.uninstall()
        for name in ("diff_surfel_rasterization", "diff_surfel_rasterization._C"):
            sys.modules.pop(name, None)


REFERENCE = "/root/reference"

_AUTOSTUB = r"""
import importlib.abc, importlib.machinery, sys, types
class _Any(types.ModuleType):
    # stands in for a third-party package this image lacks (open3d, plyfile, cv2 ...): any attribute, any submodule
    __path__ = []
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (), {"__init__": lambda self, *a, **k: None})
class _Loader(importlib.abc.Loader):
    def create_module(self, spec): return _Any(spec.name)
    def exec_module(self, module): pass
class _Last(importlib.abc.MetaPathFinder):
    MISSING = ("open3d", "cv2", "trimesh", "einsum", "lpips", "pyrender", "e3nn", "kornia", "plyfile", "matplotlib", "sklearn",
               "PIL", "mediapy", "skimage", "imageio", "umap", "torchvision", "pytorch3d", "nvdiffrast", "xatlas", "pymeshlab")
    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] not in self.MISSING:
            return None
        return importlib.machinery.ModuleSpec(name, _Loader(), is_package=True)
sys.meta_path.append(_Last())
"""


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "gaussian_renderer")), reason="reference checkout not present")
def test_real_reference_packages_resolve_to_the_library():
    """Build container only: the reference's OWN gaussian_renderer / utils.contrastive_utils modules are executed (its
    third-party imports this image lacks are auto-stubbed), from its own directory, and come out rebound."""
    code = _AUTOSTUB + textwrap.dedent("""
        import os
        sys.path.insert(0, os.getcwd())          # what `python train.py` does
        from utils.loss_utils import l1_loss, cos_loss, ssim            # train.py:15
        from gaussian_renderer import render, network_gui              # train.py:16
        from utils.contrastive_utils import *                          # train_semantic.py:9
        import gaussian_renderer, utils.contrastive_utils as cu
        import instascene_amd.render, instascene_amd.contrastive
        assert gaussian_renderer.__file__.startswith(os.getcwd()), gaussian_renderer.__file__
        assert cu.__file__.startswith(os.getcwd()), cu.__file__
        assert render is instascene_amd.render.render
        assert contrastive_loss is instascene_amd.contrastive.contrastive_loss
        assert callable(feature_to_rgb) and feature_to_rgb.__module__ == "utils.contrastive_utils"
        assert hasattr(network_gui, "try_connect") and l1_loss.__module__ == "utils.loss_utils"
        from scene.gaussian_model import distCUDA2                     # scene/gaussian_model.py:21
        import instascene_amd.knn
        assert distCUDA2 is instascene_amd.knn.distCUDA2
        print("ok")
    """)
    r = subprocess.run([sys.executable, "-c", code], cwd=REFERENCE, env=_env(os.path.join(ROOT, "dropin")),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-3000:]


def test_install_makes_empty_cache_act_only_under_memory_pressure(monkeypatch):
    """train_semantic.py:208 calls torch.cuda.empty_cache() every iteration; under the drop-in that hands memory back only when
    the device is short of it (dropin.empty_cache_under_pressure), ISR_KEEP_EMPTY_CACHE=1 opts out, uninstall() restores."""
    import torch
    from instascene_amd import dropin
    real = torch.cuda.empty_cache
    calls = []
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: calls.append(1))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    free = [200 << 30]
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda: (free[0], 288 << 30))
    try:
        dropin.install()
        assert torch.cuda.empty_cache is not real
        torch.cuda.empty_cache()
        assert calls == []                      # plenty of free memory: nothing is handed back
        free[0] = 10 << 30
        torch.cuda.empty_cache()
        assert calls == [1]                     # under pressure the real function runs
    finally:
        dropin.uninstall()
    torch.cuda.empty_cache()
    assert calls == [1, 1]                      # restored: the (patched-in) original again
    monkeypatch.setenv("ISR_KEEP_EMPTY_CACHE", "1")
    try:
        dropin.install()
        torch.cuda.empty_cache()
        assert calls == [1, 1, 1]
    finally:
        dropin.uninstall()


def test_dropin_c_module_logs_its_fallback_and_shares_one_mode(tmp_path):
    """diff_surfel_rasterization._C (advisor, round 5): when the compiled extension cannot be imported the Python binding serves the
    names - and says so once on logger `instascene_amd`; `_C.set_mode` sets the mode of BOTH bindings (rasterizer.set_mode is the one
    source of truth), and an extension that loads is handed the Python layer's current mode."""
    code = textwrap.dedent("""
        import json, logging, sys, types
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        recs = []
        class H(logging.Handler):
            def emit(self, r): recs.append(r.getMessage())
        logging.getLogger("instascene_amd").addHandler(H())
        from instascene_amd import rasterizer as rz
        rz.set_mode("fast")
        broken = %s
        if broken:          # an extension that fails to import (stale build, built against another torch)
            sys.modules["instascene_amd._C_hip"] = None
        import diff_surfel_rasterization._C as C
        C.set_mode("exact")
        out = {"compiled": C.COMPILED, "reason": C.FALLBACK_REASON, "logged": [m for m in recs if "_C_hip.so did not load" in m],
               "mode_after": rz.get_mode(), "mirror": C.rasterize_gaussians.__module__}
        print(json.dumps(out))
    """)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for broken in (True, False):
        r = subprocess.run([sys.executable, "-c", code % (root, os.path.join(root, "dropin"), broken)], capture_output=True, text=True,
                           env=_env(""), timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        assert rec["mode_after"] == "exact"
        if broken:
            assert rec["compiled"] is False and "_C_hip" in rec["reason"] and len(rec["logged"]) == 1
            assert rec["mirror"] == "instascene_amd.rasterizer"
        elif rec["compiled"]:
            assert rec["reason"] is None and rec["logged"] == []
