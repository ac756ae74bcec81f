"""CPU tests of the host-side logic: the C-ABI library exports every declared symbol, render()'s
post-processing matches the reference's own outputs, argument checking mirrors the reference."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = []
    for h in ("instascene_rasterizer.h", "instascene_ops.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(is[ro]_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_loads_and_exports_every_declared_symbol():
    from instascene_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    import torch  # noqa: F401  (one HIP runtime per process: torch's is loaded first, see _lib.lib)
    L = ctypes.CDLL(_lib.LIB_PATH)
    decl = _declared_symbols()
    assert len(decl) >= 15
    for name in decl:
        assert hasattr(L, name), f"{name} declared in include/ but not exported"
    assert set(decl) == set(_lib.SIGNATURES.keys())
    # size queries and version are host-only: callable without a GPU
    L.isr_version.restype = ctypes.c_int
    assert L.isr_version() >= 1
    L.isr_geom_bytes.restype = ctypes.c_size_t
    L.isr_image_bytes.restype = ctypes.c_size_t
    L.isr_binning_bytes.restype = ctypes.c_size_t
    L.isr_binning_bytes.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    assert L.isr_geom_bytes(1000) > 1000 * 80
    assert L.isr_image_bytes(64, 48) > 64 * 48 * 20
    assert L.isr_binning_bytes(1000, 64, 48) >= 12 * 1000


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "instascene_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle", src, flags=re.M), f
                assert "surfel_oracle" not in src or f.endswith((".hip", ".hpp")), f
    for f in ("diff_surfel_rasterization/__init__.py", "diff_surfel_rasterization/_C.py", "simple_knn/_C.py", "sitecustomize.py"):
        assert "oracle" not in open(os.path.join(ROOT, "dropin", f)).read()


def test_render_post_processing_has_no_host_path(golden_dir):
    """post_process is two HIP kernels each way; handing it host tensors must fail loudly instead of computing on the CPU
    (the CPU restatement pinned by tests/golden/render_post.npz is oracle/torch_ops.render_post, test_goldens_oracle.py;
    the HIP kernels are pinned by the same fixture in tests/test_gpu_ops.py)."""
    from instascene_amd.render import post_process
    from instascene_amd import scenes
    c = np.load(os.path.join(golden_dir, "cameras.npz"))
    W, H = (int(v) for v in c["wh0"])
    cam = scenes.Camera(W, H, float(c["fov0"][0]), float(c["fov0"][1]), torch.tensor(c["wvt0"]),
                        torch.tensor(c["proj0"]), torch.tensor(c["full0"]), torch.tensor(c["center0"]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        post_process(cam, torch.zeros(7, H, W), 1.0)


def test_camera_ray_table_matches_the_reference_points(golden_dir):
    """render._camera_rays (one 3x3 matrix, broadcast) against the reference's depth_to_normal fixture: the normals of
    depth * rays_d + rays_o equal tests/golden/depth_to_normal.npz."""
    from instascene_amd.render import _camera_rays
    from instascene_amd import scenes
    from helpers import assert_close
    z = np.load(os.path.join(golden_dir, "depth_to_normal.npz"))
    c = np.load(os.path.join(golden_dir, "cameras.npz"))
    for i in range(2):
        W, H = (int(v) for v in c[f"wh{i}"])
        cam = scenes.Camera(W, H, float(c[f"fov{i}"][0]), float(c[f"fov{i}"][1]), torch.tensor(c[f"wvt{i}"]),
                            torch.tensor(c[f"proj{i}"]), torch.tensor(c[f"full{i}"]), torch.tensor(c[f"center{i}"]))
        rays_d, rays_o = _camera_rays(cam, "cpu")
        pts = (torch.tensor(z[f"depth{i}"]).reshape(-1, 1) * rays_d + rays_o).reshape(H, W, 3)
        n = torch.zeros_like(pts)
        n[1:-1, 1:-1] = torch.nn.functional.normalize(
            torch.linalg.cross(pts[2:, 1:-1] - pts[:-2, 1:-1], pts[1:-1, 2:] - pts[1:-1, :-2], dim=-1), dim=-1)
        assert_close(n.numpy(), z[f"normal{i}"], 2e-5, "normal from the ray table")


def test_rasterizer_argument_checks_mirror_reference():
    from instascene_amd.rasterizer import GaussianRasterizer, GaussianRasterizationSettings
    s = GaussianRasterizationSettings(8, 8, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3),
                                      False, False)
    r = GaussianRasterizer(s)
    x = torch.zeros(4, 3)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(x, x, torch.zeros(4, 1), shs=None, colors_precomp=None, scales=torch.zeros(4, 2), rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed"):
        r(x, x, torch.zeros(4, 1), colors_precomp=x, scales=None, rotations=None, cov3D_precomp=None)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):       # reference CHECK_INPUT
        r(x, x, torch.zeros(4, 1), colors_precomp=x, scales=torch.zeros(4, 2), rotations=torch.zeros(4, 4))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from instascene_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_render_package_evaluates_derived_maps_on_first_access(golden_dir, monkeypatch):
    """The seven allmap-derived entries of render()'s dict are computed on first access, under the grad mode of the
    render() call, and dict(pkg) / items() see real tensors.  (Host tensors: the kernels are replaced by the oracle's
    restatement for this test of the dict's laziness.)"""
    from instascene_amd import render as R
    from instascene_amd import scenes
    from oracle import torch_ops

    def _post(cam, allmap, ratio):
        return torch_ops.render_post(allmap, cam.world_view_transform, cam.full_proj_transform, cam.image_width,
                                     cam.image_height, ratio)
    monkeypatch.setattr(R, "post_process", _post)
    z = np.load(os.path.join(golden_dir, "render_post.npz"))
    c = np.load(os.path.join(golden_dir, "cameras.npz"))
    W, H = (int(v) for v in c["wh0"])
    cam = scenes.Camera(W, H, float(c["fov0"][0]), float(c["fov0"][1]), torch.tensor(c["wvt0"]),
                        torch.tensor(c["proj0"]), torch.tensor(c["full0"]), torch.tensor(c["center0"]))
    allmap = torch.tensor(z["c0_r0_allmap"]).requires_grad_(True)
    pkg = R.RenderPackage({"render": torch.zeros(3, 4, 4)})
    dict.update(pkg, dict.fromkeys(R._LAZY_KEYS))
    pkg._pending = (cam, allmap, 0.0, True)
    assert set(pkg.keys()) == {"render", *R._LAZY_KEYS}
    with torch.no_grad():                       # accessed later under no_grad: still differentiable
        a = pkg["rend_alpha"]
    assert pkg._pending is None and a.requires_grad
    want = _post(cam, allmap, 0.0)
    for k in R._LAZY_KEYS:
        assert torch.equal(pkg[k], want[k])
    pkg2 = R.RenderPackage({"render": torch.zeros(3, 4, 4)})
    dict.update(pkg2, dict.fromkeys(R._LAZY_KEYS))
    pkg2._pending = (cam, allmap, 0.0, False)
    assert all(v is not None for v in dict(pkg2).values()) and not pkg2["surf_normal"].requires_grad
    assert all(v is not None for _, v in pkg2.items())


def test_async_capacity_rounding_is_monotone_and_tight():
    """rasterizer._round_capacity: never below the request, at most 1/32 above it, monotone, and few distinct values over the
    spread of a scene's views (what keeps the allocator's pools from growing view by view)."""
    from instascene_amd.rasterizer import _round_capacity
    prev = 0
    for r in list(range(0, 5000, 7)) + [10 ** 5 + k * 977 for k in range(200)] + [19_400_000 + k * 3001 for k in range(200)]:
        c = _round_capacity(r)
        assert c >= r and (r == 0 or c <= r + max(1, r // 32) + 1), (r, c)
    vals = [_round_capacity(r) for r in range(19_300_000, 19_500_000, 1000)]
    assert vals == sorted(vals) and len(set(vals)) <= 2
    assert _round_capacity(0) == 0 and _round_capacity(-5) == -5


def test_debug_flag_dumps_the_argument_snapshot_on_failure(tmp_path, monkeypatch):
    """``raster_settings.debug=True`` (reference diff_surfel_rasterization/__init__.py:93-101,150-158): when the native call
    raises, the wrapper writes the CPU copy of its arguments taken BEFORE the call to snapshot_fw.dump / snapshot_bw.dump
    and re-raises.  No GPU needed: the native entry points are replaced by ones that fail."""
    import torch
    from instascene_amd import rasterizer as rz

    class FakeLib:
        calls = []

        def isr_set_debug(self, on, fault_after):
            FakeLib.calls.append((on, fault_after))
            return 0

    monkeypatch.setattr(rz, "lib", lambda: FakeLib())
    monkeypatch.chdir(tmp_path)
    P = 5
    means3D = torch.randn(P, 3, requires_grad=True)
    means2D = torch.zeros(P, 3, requires_grad=True)
    e = torch.empty(0)
    settings = rz.GaussianRasterizationSettings(image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3),
                                                scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0,
                                                campos=torch.zeros(3), prefiltered=False, debug=True)

    def boom(*a, **k):
        a[1].mul_(0)             # "corrupt" an input after the snapshot was taken
        raise rz._lib.IsrError("isr_forward_render failed (code -2): [debug] kernel k_render_fwd failed: injected")

    monkeypatch.setattr(rz, "rasterize_gaussians", boom)
    with pytest.raises(rz._lib.IsrError, match="k_render_fwd"):
        with torch.no_grad():
            rz._RasterizeGaussians.apply(means3D.detach().clone(), means2D, e, torch.rand(P, 3), torch.rand(P, 1), torch.rand(P, 2),
                                         torch.rand(P, 4), e, e, settings)
    snap = torch.load(tmp_path / "snapshot_fw.dump")
    assert len(snap) == 21 and snap[-1] is True and snap[14] == 8          # the _C.rasterize_gaussians argument tuple
    assert float(snap[1].abs().sum()) > 0                                    # copied before the call could touch it
    assert FakeLib.calls == [(1, 0), (0, 0)]                                 # debug mode on around the call, restored after

    # backward: a context as the forward would have left it
    class Ctx:
        raster_settings = settings
        num_rendered = 3
        mode = 0
        sample_pixels = None
        needs_input_grad = (True,) * 9 + (False,) * 4
        saved_tensors = (e, means3D.detach(), torch.rand(P, 2), torch.rand(P, 4), e, torch.ones(P, dtype=torch.int32), e,
                         torch.rand(P, 1, 3), torch.zeros(4, dtype=torch.uint8), torch.zeros(4, dtype=torch.uint8),
                         torch.zeros(4, dtype=torch.uint8))

    def boom_b(*a, **k):
        raise rz._lib.IsrError("isr_backward failed (code -2): [debug] kernel k_render_bwd failed: injected")

    monkeypatch.setattr(rz, "rasterize_gaussians_backward", boom_b)
    FakeLib.calls.clear()
    with pytest.raises(rz._lib.IsrError, match="k_render_bwd"):
        rz._RasterizeGaussians.backward(Ctx, torch.ones(3, 8, 8), None, torch.ones(7, 8, 8), None, None)
    snap = torch.load(tmp_path / "snapshot_bw.dump")
    assert len(snap) == 24 and snap[20] == 3 and snap[-1] is True          # the _C.rasterize_gaussians_backward argument tuple
    assert FakeLib.calls == [(1, 0), (0, 0)]


def test_workspace_size_classes_are_geometric():
    """rasterizer._size_class: the smallest member >= n of a x1.25 sequence - a drifting count changes the requested
    workspace size only every +25 %."""
    from instascene_amd.rasterizer import _size_class
    seen = set()
    prev = 0
    for n in [1, 4096, 4097, 10_000, 611_750, 720_893, 5_870_199, 15_467_272, 3 * 10 ** 9]:
        c = _size_class(n)
        assert c >= n and c >= prev and (n <= 4096 or c < 1.26 * n + 256)
        assert _size_class(c) == c
        prev = c
        seen.add(c)
    drift = {_size_class(n) for n in range(611_750, 720_893, 997)}      # the C2 soak's 18 % drift of R
    assert len(drift) <= 2


def test_trainer_defaults_are_the_reference_defaults(golden_dir):
    """tests/golden/defaults.npz = arguments/__init__.py:89-127 read by importing the reference's OptimizationParams: the
    harness trainers' default hyper-parameters (loss weights, learning rates, density-control schedule) are those."""
    import inspect
    import numpy as np
    from instascene_amd import harness
    z = {k: float(v) for k, v in np.load(os.path.join(golden_dir, "defaults.npz")).items()}
    seg = {k: v.default for k, v in inspect.signature(harness.SegTrainer.__init__).parameters.items()}
    assert (seg["lambda_sv"], seg["lambda_mv"], seg["lambda_3d"], seg["sample_mv_frames"]) == (
        z["lambda_singview_contras"], z["lambda_multiview_contras"], z["lambda_3D_contras"], int(z["sample_mv_frames"]))
    plain = {k: v.default for k, v in inspect.signature(harness.PlainSegTrainer.__init__).parameters.items()}
    assert (plain["lambda_sv"], plain["lambda_mv"], plain["lambda_3d"], plain["sample_mv_frames"]) == (
        seg["lambda_sv"], seg["lambda_mv"], seg["lambda_3d"], seg["sample_mv_frames"])
    rgb = {k: v.default for k, v in inspect.signature(harness.RgbTrainer.__init__).parameters.items()}
    assert (rgb["lambda_dssim"], rgb["lambda_normal"], rgb["lambda_dist"]) == (z["lambda_dssim"], z["lambda_normal"], z["lambda_dist"])
    src = inspect.getsource(harness)
    assert "lr=0.025" in src and z["seg_feature_lr"] == 0.025
    lrs = {g["name"]: g["lr"] for g in harness.RgbGaussianModel.param_groups(type("M", (), dict.fromkeys(
        ["_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"]))())}
    assert lrs == {"xyz": z["position_lr_init"], "f_dc": z["feature_lr"], "f_rest": z["feature_lr"] / 20.0, "opacity": z["opacity_lr"],
                   "scaling": z["scaling_lr"], "rotation": z["rotation_lr"]}
    dens = dict(from_iter=int(z["densify_from_iter"]), until_iter=int(z["densify_until_iter"]), interval=int(z["densification_interval"]),
                opacity_reset_interval=int(z["opacity_reset_interval"]), grad_threshold=z["densify_grad_threshold"],
                opacity_cull=z["opacity_cull"], percent_dense=z["percent_dense"])
    for k, v in dens.items():
        assert f"{k}={v:_}" in src.replace("15_000", "15_000") or f"{k}={v}" in src, (k, v)


def test_arena_lease_ends_with_the_last_view_of_the_block():
    """instascene_amd/arena.py: a block is free again exactly when every tensor viewing its storage has died - views of views
    and tensors saved by an autograd graph included - read from the storage's reference count (no hook, no explicit release)."""
    import torch
    from instascene_amd import arena
    b = arena._Block.__new__(arena._Block)
    b._adopt(torch.empty(4096, dtype=torch.uint8), 4096)
    assert b.idle()
    v = b.base[:100].view(torch.float32).view(5, 5)
    assert not b.idle()
    w = v.t()
    del v
    assert not b.idle()                 # a view of the view keeps the lease
    del w
    assert b.idle()
    x = torch.ones(5, 5, requires_grad=True)
    v = b.base[:100].view(torch.float32).view(5, 5)
    y = (x * v).sum()                   # the graph saves v for the backward
    del v
    assert not b.idle()
    y.backward()
    del y
    assert b.idle()
    # size classes grow by x1.25 and requests below MIN_BYTES / on the CPU are plain torch.empty
    assert arena._class_of(arena.MIN_BYTES + 1) <= int(arena.MIN_BYTES * 1.26)
    t = arena.empty((3, 4), torch.float32, "cpu")
    assert t.shape == (3, 4) and arena.reserved_bytes() == 0


def test_arena_leases_are_not_autograd_views():
    """A lease is a tensor of its own on the block's storage (Tensor.set_), not a view of the block: an output of a custom
    autograd.Function built that way takes in-place operations like the reference's fresh tensors do (round 4 handed out
    `base[:n].view(dtype).view(shape)`, on which `image.clamp_()` raised "is a view and is being modified inplace"), and the
    storage's reference count still ends the lease with the last tensor on it."""
    import torch
    from instascene_amd import arena
    b = arena._Block.__new__(arena._Block)
    b._adopt(torch.empty(4096, dtype=torch.uint8), 4096)

    def lease(shape):
        return torch.empty(0, dtype=torch.float32).set_(b._st, 0, shape)

    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            out = lease((3, 4))
            out.copy_(x * 2)
            return out

        @staticmethod
        def backward(ctx, g):
            return g * 2

    x = torch.ones(3, 4, requires_grad=True)
    y = F.apply(x)
    assert not b.idle() and y._base is None and y.requires_grad
    y.clamp_(0, 1)                       # in-place on the output
    y[1:2].mul_(3)                       # ... and on a slice of it
    y.sum().backward()
    assert x.grad is not None
    del y
    assert b.idle()
    # the round-4 form, for the record: it raises
    class G(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            out = b.base[:48].view(torch.float32).view(3, 4)
            out.copy_(x * 2)
            return out

        @staticmethod
        def backward(ctx, g):
            return g * 2
    z = G.apply(x)
    import pytest
    with pytest.raises(RuntimeError):
        z.clamp_(0, 1)


def test_arena_trim_asks_the_pools_own_stream():
    """trim() must judge a block by the stream that OWNS its pool (the pool key), not by the caller's current stream: a block
    used on a foreign stream would otherwise count as free the moment trim runs on that stream (advisor finding, round 4)."""
    import torch
    from instascene_amd import arena

    class FakeBlock:
        nbytes = 4096

        def __init__(self):
            self.asked = []
            self.base = torch.empty(1)

        def reusable(self, stream):
            self.asked.append(stream)
            return False

    fb = FakeBlock()
    key = (0, 0xABCDEF, 1 << 20)
    arena._POOLS[key] = [fb]
    try:
        assert arena.trim() == 0
        assert fb.asked == [0xABCDEF]
    finally:
        arena._POOLS.pop(key, None)


def test_arena_block_created_after_a_trim_stays_in_its_pool(monkeypatch):
    """Advisor finding, round 5: _lease() holds its pool's list while trim() runs (cap path, out-of-memory retry); trim() used to
    rebind / delete that list, so the new block was appended to a dead list - never reused, never trimmed, counted forever.
    Lease past a 3 MiB cap on the CPU and check that every live block is in a pool and the total stays bounded."""
    import torch
    from instascene_amd import arena, _hot
    monkeypatch.setattr(_hot, "raw_stream", lambda dev: 0)
    monkeypatch.setattr(arena, "MAX_BYTES", 3 << 20)
    monkeypatch.setattr(arena, "_POOLS", {})
    monkeypatch.setattr(arena, "_BY_PTR", {})
    monkeypatch.setattr(arena, "_TOTAL", [0])
    dev = torch.device("cpu", 0)
    n = arena.MIN_BYTES
    for i in range(12):
        t = arena._lease((n,), torch.uint8, dev, n)      # the previous lease died: its block is idle when the next one asks
        u = arena._lease((n,), torch.uint8, dev, n)
        v = arena._lease((n,), torch.uint8, dev, n)
        w = arena._lease((n,), torch.uint8, dev, n)      # the fourth live block: past the cap, trim() runs inside _lease
        in_pools = {id(b) for pool in arena._POOLS.values() for b in pool}
        assert {id(b) for b in arena._BY_PTR.values()} == in_pools, "a block is in no pool"
        assert arena.reserved_bytes() == sum(b.nbytes for pool in arena._POOLS.values() for b in pool)
        assert arena.reserved_bytes() <= 4 * arena._class_of(n)
        del t, u, v, w
    # and the out-of-memory retry path: trim() from inside _Block.__init__
    real_empty, calls = torch.empty, [0]

    def flaky_empty(*a, **k):
        if k.get("dtype") == torch.uint8 and a and a[0] >= n:
            calls[0] += 1
            if calls[0] == 1:
                raise torch.OutOfMemoryError("simulated")
        return real_empty(*a, **k)
    monkeypatch.setattr(arena, "_real_empty_cache", lambda: None)
    keep = [arena._lease((n,), torch.uint8, dev, n) for _ in range(2)]
    monkeypatch.setattr(torch, "empty", flaky_empty)
    extra = arena._lease((2 * n,), torch.uint8, dev, 2 * n)     # a new size class: needs a new block, whose first attempt "fails"
    monkeypatch.setattr(torch, "empty", real_empty)
    assert calls[0] >= 2
    in_pools = {id(b) for pool in arena._POOLS.values() for b in pool}
    assert {id(b) for b in arena._BY_PTR.values()} == in_pools
    del keep, extra


def test_bench_final_line_is_compact_and_round_trips():
    """bench.py prints ONE compact JSON line (round 5's 24.8 KB line was lost by the driver's parser): from a canned full record of
    the size bench.py really produces, the line stays below 8 KB, round-trips through json, and carries the contract keys, the
    dominant kernel's roofline and the CPU baseline."""
    import json
    import bench
    kern = {("k_%02d" % i): {"ms_per_launch": 0.1234, "launches_per_view": 1.0, "algorithmic_bytes": 1 << 30, "GB/s": 1234.5,
                               "frac_hbm": 0.1543} for i in range(40)}
    sub = {"value": 123.456, "ms_per_step": 8.1, "note": "n" * 600, "kernels_ms_x_launches_per_step": {k: [0.1, 1.0] for k in kern}}
    full = {"metric": "train-step views/sec (fwd+bwd) @1.5M Gaussians, 1080p, 32-d feat", "value": 618.01, "unit": "views/s",
            "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 1.6181, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "unmodified_driver": {"views_per_s": 63.7, "workspace": "w" * 300},
            "config": {"workload": "C3: " + "w" * 400, "parallelism": "dp1 (one view per rank)", "arithmetic_mode": "fast_reflists",
                       "integer_state": "the reference's, bit for bit", "view_order": "v" * 900,
                       "hoisted_out_of_the_timed_region": ["h" * 300] * 5, "rccl_world_size": 1, "tracer": True,
                       "views_per_s_exact": 400.7, "views_per_s_C2_rgb": 1307.4, "views_per_s_C5_seg": 247.9,
                       "views_per_s_unmodified_driver": 63.7, "views_per_s_reference_defaults": 404.2,
                       "gradient_exchange": {"kind": "RCCL all-reduce", "bytes": {"buffer": 192000000}}},
            "roofline": {"bound": "hbm", "kernel": "k_render_fwd", "achieved": 1748.05, "peak": 8000.0, "unit": "GB/s", "frac": 0.2185,
                         "frac_hbm": 0.2185, "frac_fp32_flops": 0.1886, "avg_launch_ms": 0.9088, "launches_per_step": 1.0,
                         "traffic": 1545011200, "traffic_stale": False, "traffic_over_algorithmic_bytes": 0.973,
                         "hbm": {"bytes": 1588683804}, "note": "x" * 500, "kernels": kern,
                         "workload": {"P": 1500000, "V": 1280051, "R": 5876701, "N": 2073600, "F": 32, "tiles": 8160},
                         "valu": {"pixel_splat_pairs_evaluated": 445911232, "pixel_splat_pairs_contributing": 108699966,
                                  "flops": 26967246424, "lane_utilisation_of_blending_pairs": 0.33, "flop_model": "m" * 200},
                         "issue": {"issue_frac": 0.49, "model": "m" * 400}},
            "sub_records": {("s%d" % i): sub for i in range(14)},
            "cpu_baseline": {"value": 0.185, "unit": "views/s", "cores": 128, "cpu": "AMD EPYC 9575F 64-Core Processor", "kind": "port",
                             "build": "g++ -O3 -march=native", "sample": "s" * 700}}
    assert len(json.dumps(full)) > 20000                      # the canned record is as big as the real one
    line = json.dumps(bench.compact_record(full), separators=(",", ":"))
    assert len(line) < 8192 and len(line) <= bench.COMPACT_LIMIT and "\n" not in line
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert back["value"] == 618.01 and back["config"]["views_per_s_C2_rgb"] == 1307.4 and back["config"]["workload"].startswith("C3")
    r = back["roofline"]
    assert r["bound"] == "hbm" and r["frac"] == 0.2185 and r["bytes"] == 1588683804 and r["R"] == 5876701 and r["traffic"] == 1545011200
    assert abs(r["bytes"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / r["peak"] - r["frac"]) < 2e-3      # frac is recomputable from the line
    assert back["cpu_baseline"]["cores"] == 128 and back["cpu_baseline"]["kind"] == "port"
    assert "sub_records" not in back and back["details"] == "bench_details.json"
    # a multi-rank record: cpu_baseline null, the exchange's bytes kept
    full["cpu_baseline"] = None
    back = json.loads(json.dumps(bench.compact_record(full)))
    assert back["cpu_baseline"] is None and back["config"]["gradient_exchange"]["bytes"]["buffer"] == 192000000
