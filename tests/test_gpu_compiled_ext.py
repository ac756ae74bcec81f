"""The compiled torch extension at the boundary (instascene_amd/_C_hip.so = `diff_surfel_rasterization._C`, reference
ext.cpp:15-18, rasterize_points.cu:39-295): the reference's three entry points with the reference's argument order and return
tuples, called the way the reference's own autograd wrapper calls them (diff_surfel_rasterization/__init__.py:96-113,153-176,
203-208), against the ctypes host layer over the same C ABI and against the CPU oracle."""
import math

import numpy as np
import pytest
import torch

from helpers import small_scene, oracle_forward

pytestmark = pytest.mark.gpu


def _args(inp, cam, F, W, H):
    e = torch.empty(0, device="cuda")
    return (torch.zeros(3, device="cuda"), inp["means3D"].cuda(), e, inp["opacities"].cuda(), inp["scales"].cuda(),
            inp["rotations"].cuda(), 1.0, e, inp["extra"].cuda() if F else e, F, cam.world_view_transform.cuda(),
            cam.full_proj_transform.cuda(), math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), H, W, inp["shs"].cuda(), 3,
            cam.camera_center.cuda(), False, False)


@pytest.mark.parametrize("mode", ["exact", "fast_reflists", "fast"])
def test_compiled_extension_equals_the_ctypes_layer_and_the_oracle(mode):
    from instascene_amd import _C_hip, rasterizer as rz
    from instascene_amd._lib import GRAD_EXTRA, GRAD_GEOMETRY, MODE_EXACT, MODE_FAST
    sc, cams, inp = small_scene(P=1500, F=8, W=96, H=64, seed=41)
    cam, F, W, H = cams[1], 8, 96, 64
    a = _args(inp, cam, F, W, H)
    rz.set_mode(mode)                                   # reaches the extension too
    try:
        c = _C_hip.rasterize_gaussians(*a)
        assert len(c) == 10
        R, color, others, radii, extra, geom, binning, img, pairs, last = c
        py = rz.rasterize_gaussians(*a, tracer=True)
        assert R == py[0]
        for x, y, name in ((color, py[1], "color"), (others, py[2], "others"), (radii, py[3], "radii"), (extra, py[4], "extra")):
            assert torch.equal(x, y), name
        n = int(last.item()) + 1
        assert n == int(py[9].item()) + 1
        key = lambda t: sorted(map(tuple, t[:n].cpu().tolist()))
        assert key(pairs) == key(py[8])                 # the tracer list: the same pairs (order is unspecified, as in the reference)
        if mode == "exact":
            st = oracle_forward(inp, cam)
            assert R == st["R"]
            np.testing.assert_array_equal(color.cpu().numpy(), st["color"])
            np.testing.assert_array_equal(extra.cpu().numpy(), st["extra"])
        g = torch.Generator().manual_seed(1)
        dC, dO, dE = (torch.randn(s, generator=g).cuda() for s in ((3, H, W), (7, H, W), (F, H, W)))
        e = torch.empty(0, device="cuda")
        # the reference wrapper's call (diff_surfel_rasterization/__init__.py:129-151)
        got = _C_hip.rasterize_gaussians_backward(a[0], a[1], radii, e, a[4], a[5], a[8], 1.0, e, a[10], a[11], a[12], a[13], dC, dO, dE,
                                                  a[16], 3, a[18], geom, R, binning, img, False)
        want = rz.rasterize_gaussians_backward(a[0], a[1], py[3], e, a[4], a[5], a[8], 1.0, e, a[10], a[11], a[12], a[13], dC, dO, dE,
                                               a[16], 3, a[18], py[5], py[0], py[6], py[7], False, grad_mask=GRAD_EXTRA | GRAD_GEOMETRY,
                                               mode=MODE_EXACT if mode == "exact" else MODE_FAST)
        assert len(got) == 9
        for i, (x, y) in enumerate(zip(got, want)):
            assert torch.equal(x.reshape(y.shape), y), i
        vis = _C_hip.mark_visible(a[1], a[10], a[11])
        assert vis.dtype == torch.bool and torch.equal(vis, rz.mark_visible(a[1], a[10], a[11]))
    finally:
        rz.set_mode("exact")


def test_dropin_module_serves_the_compiled_extension():
    """`import diff_surfel_rasterization._C` under the drop-in resolves to the compiled module, and the empty scene behaves like
    the reference's (colour = background, zero maps, no pairs)."""
    import sys
    from instascene_amd import dropin
    dropin.install()
    try:
        sys.modules.pop("diff_surfel_rasterization._C", None)
        import diff_surfel_rasterization._C as C
        assert C.COMPILED and C.rasterize_gaussians.__module__ != "instascene_amd.rasterizer"
        e = torch.empty(0, device="cuda")
        cam = small_scene(P=10, F=0, W=32, H=32)[1][0]
        out = C.rasterize_gaussians(torch.tensor([0.2, 0.3, 0.4], device="cuda"), torch.empty(0, 3, device="cuda"), e, e, e, e, 1.0, e, e, 0,
                                    cam.world_view_transform.cuda(), cam.full_proj_transform.cuda(), 1.0, 1.0, 32, 32,
                                    torch.empty(0, 16, 3, device="cuda"), 3, cam.camera_center.cuda(), False, False)
        assert out[0] == 0 and float(out[1][1].min()) == float(out[1][1].max()) == pytest.approx(0.3) and int(out[9].item()) == -1
    finally:
        dropin.uninstall()
