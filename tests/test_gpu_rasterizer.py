"""GPU parity: the HIP rasterizer (through the C-ABI library) vs the CPU oracle on the
same seeded inputs.  EXACT mode must be bit-identical on every forward output and
on all integer state; gradients within 1e-3 (relative to the tensor's max)."""
import math

import numpy as np
import pytest
import torch

import oracle
from helpers import small_scene, oracle_forward, assert_close

pytestmark = pytest.mark.gpu

gpu = pytest.importorskip("torch").cuda.is_available()
if gpu:
    from instascene_amd import rasterizer as rz, scenes
    from instascene_amd._lib import GRAD_EXTRA, GRAD_GEOMETRY, MODE_EXACT, MODE_FAST


def _dev(t):
    return None if t is None else t.cuda()


def hip_forward(inp, cam, bg=(0.0, 0.0, 0.0), sh_degree=3, mode=None, tracer=False, scale_modifier=1.0,
                colors_precomp=None, transMat_precomp=None, use_sh=True, use_extra=True, tight=False):
    dev = "cuda"
    e = lambda: torch.empty(0, device=dev)
    extra = _dev(inp["extra"]) if (use_extra and inp.get("extra") is not None) else e()
    F = extra.shape[1] if extra.numel() else 0
    sh = _dev(inp["shs"]) if (use_sh and colors_precomp is None) else e()
    col = e() if colors_precomp is None else torch.as_tensor(colors_precomp).cuda()
    tm = e() if transMat_precomp is None else torch.as_tensor(transMat_precomp).cuda()
    sc = e() if transMat_precomp is not None else _dev(inp["scales"])
    ro = e() if transMat_precomp is not None else _dev(inp["rotations"])
    args = dict(bg=torch.tensor(bg, dtype=torch.float32, device=dev), means3D=_dev(inp["means3D"]), colors=col,
                opacity=_dev(inp["opacities"]), scales=sc, rotations=ro, scale_modifier=scale_modifier,
                transMat_precomp=tm, extra_attrs=extra, attr_degree=F, viewmatrix=cam.world_view_transform.cuda(),
                projmatrix=cam.full_proj_transform.cuda(), tan_fovx=math.tan(cam.FoVx * 0.5),
                tan_fovy=math.tan(cam.FoVy * 0.5), image_height=cam.image_height, image_width=cam.image_width, sh=sh,
                degree=sh_degree, campos=cam.camera_center.cuda(), prefiltered=False, debug=False)
    out = rz.rasterize_gaussians(**args, tracer=tracer, mode=mode, tight=tight)
    return args, out


def check_forward_exact(st, args, out, tracer=False):
    R, color, others, radii, extra, geom, binning, img, grp, gidx = out
    P, W, H = st["P"], st["W"], st["H"]
    assert R == st["R"]
    np.testing.assert_array_equal(radii.cpu().numpy(), st["radii"])
    dbg = rz.debug_state(P, W, H, R, geom, binning, img)
    np.testing.assert_array_equal(dbg["tiles_touched"], st["tiles_touched"])
    np.testing.assert_array_equal(dbg["point_list"], st["point_list"])
    np.testing.assert_array_equal(dbg["ranges"], st["ranges"])
    np.testing.assert_array_equal(dbg["n_contrib"], st["n_contrib"])
    np.testing.assert_array_equal(dbg["final_T"], st["final_T"])
    np.testing.assert_array_equal(color.cpu().numpy(), st["color"])
    np.testing.assert_array_equal(others.cpu().numpy(), st["others"])
    if st["ED"]:
        np.testing.assert_array_equal(extra.cpu().numpy(), st["extra"])
    if tracer:
        n = int(gidx.item()) + 1
        got = {(int(a), int(b)) for a, b in grp[:n].cpu().numpy()}
        want = {(int(a), int(b)) for a, b in st["tracer"]}
        assert n == len(st["tracer"]) and got == want


def check_binning_exact(st, out):
    """Integer state of the geometry pass and the binning: radii, tiles_touched, point_list, ranges - bit-identical to
    the oracle's (reference rasterizer_impl.cu:70-138, auxiliary.h:68-78) in EXACT and in FAST mode alike."""
    R, radii, geom, binning, img = out[0], out[3], out[5], out[6], out[7]
    assert R == st["R"]
    np.testing.assert_array_equal(radii.cpu().numpy(), st["radii"])
    dbg = rz.debug_state(st["P"], st["W"], st["H"], R, geom, binning, img)
    np.testing.assert_array_equal(dbg["tiles_touched"], st["tiles_touched"])
    np.testing.assert_array_equal(dbg["point_list"], st["point_list"])
    np.testing.assert_array_equal(dbg["ranges"], st["ranges"])
    return dbg


@pytest.mark.parametrize("P,F,W,H,seed,bg", [
    (400, 6, 64, 48, 3, (0.0, 0.0, 0.0)),
    (1500, 16, 100, 70, 4, (0.2, 0.4, 0.9)),        # W,H not multiples of 16
    (800, 32, 96, 64, 5, (1.0, 1.0, 1.0)),
    (600, 40, 80, 48, 6, (0.0, 0.0, 0.0)),          # F > 32: two feature passes
    (700, 0, 64, 64, 7, (0.1, 0.1, 0.1)),           # no feature channel
    (3000, 8, 160, 112, 8, (0.0, 0.0, 0.0)),
])
def test_forward_exact_mode_is_bit_identical_to_oracle(P, F, W, H, seed, bg):
    sc, cams, inp = small_scene(P=P, F=F, W=W, H=H, seed=seed, mu_s=math.log(0.05))
    for cam in cams[:2]:
        st = oracle_forward(inp, cam, bg=bg, tracer=True)
        args, out = hip_forward(inp, cam, bg=bg, mode=MODE_EXACT, tracer=True)
        check_forward_exact(st, args, out, tracer=True)


def test_forward_large_splats_and_long_tile_lists():
    # big, dense splats: tile lists of several thousand entries exercise the multi-batch blend
    # and the bitonic sort beyond one pass
    sc, cams, inp = small_scene(P=6000, F=4, W=64, H=64, seed=11, mu_s=math.log(0.25))
    st = oracle_forward(inp, cams[0])
    lens = st["ranges"][:, 1].astype(np.int64) - st["ranges"][:, 0]
    assert lens.max() > 2000
    args, out = hip_forward(inp, cams[0], mode=MODE_EXACT)
    check_forward_exact(st, args, out)


@pytest.mark.parametrize("P,lo,hi", [(9000, 4096, 16384), (60000, 16384, 10 ** 9)])
def test_forward_bucket_larger_than_lds_sort_budget(P, lo, hi):
    """Tile lists beyond the 4 096-key LDS sort: 4 096 < n <= 16 384 go to k_tile_sort_big (128 KB of LDS), longer ones to
    the in-place network in global memory; both orders are the oracle's bit for bit."""
    sc, cams, inp = small_scene(P=P, F=0, W=32, H=32, seed=12, mu_s=math.log(0.4))
    st = oracle_forward(inp, cams[0])
    lens = st["ranges"][:, 1].astype(np.int64) - st["ranges"][:, 0]
    assert lo < lens.max() <= hi
    args, out = hip_forward(inp, cams[0], mode=MODE_EXACT)
    check_forward_exact(st, args, out)


def test_radix_sorted_buckets_with_depth_ties():
    """4 096 < n <= 8 192 keys: k_tile_sort_radix (LDS radix passes on the depth bits, then equal depths into id order).  A
    tenth of the splats are exact copies of others' positions - runs of two and three equal depths in every list - and the order
    is still the oracle's (= the reference's stable sort over keys emitted in id order) bit for bit."""
    sc, cams, inp = small_scene(P=6500, F=0, W=32, H=32, seed=14, mu_s=math.log(0.4))
    inp = dict(inp)
    xyz = inp["means3D"].clone()
    xyz[3000:3400] = xyz[100:500]
    xyz[5000:5200] = xyz[100:300]           # triples
    inp["means3D"] = xyz
    st = oracle_forward(inp, cams[0])
    lens = st["ranges"][:, 1].astype(np.int64) - st["ranges"][:, 0]
    assert 4096 < lens.max() <= 8192
    args, out = hip_forward(inp, cams[0], mode=MODE_EXACT)
    check_forward_exact(st, args, out)


def test_tile_counter_aggregation_overflow_path():
    """K1 and the key scatter merge a workgroup's tile counters in a 2048-entry LDS hash table; a workgroup that touches
    more distinct tiles than fit (splats covering a 4800-tile image) must fall back to direct global atomics for the
    rest, in both kernels alike: tile lists stay bit-identical to the oracle's."""
    sc, cams, inp = small_scene(P=700, F=0, W=1280, H=960, seed=31, mu_s=math.log(0.02))
    inp = dict(inp)
    scales = inp["scales"].clone()
    scales[5:9] = 1.0                      # four splats over most of the 4800-tile image, all in the first workgroup
    scales[300:302] = 1.0                  # and two in the second
    inp["scales"] = scales
    st = oracle_forward(inp, cams[0])
    assert int((st["tiles_touched"] >= 3000).sum()) >= 5
    for mode, tight in ((MODE_EXACT, False), (MODE_FAST, False), (MODE_FAST, True)):
        args, out = hip_forward(inp, cams[0], mode=mode, tight=tight)
        if mode == MODE_EXACT:
            check_forward_exact(st, args, out)
        else:       # FAST: rcp/exp arithmetic (and optionally tight tile rectangles); isolated threshold-flip pixels allowed
            if not tight:
                check_binning_exact(st, out)
            err = np.abs(out[1].cpu().numpy() - st["color"]).max(axis=0)
            assert (err <= 1e-4 * np.abs(st["color"]).max()).mean() > 0.999 and err.max() < 0.05


def test_spatial_sort_is_a_pure_relabelling():
    """scenes.spatially_sorted stores the same Gaussians in Z-order: images are bit-identical, per-Gaussian outputs and
    gradients are the same rows in the new order."""
    sc, cams, inp = small_scene(P=3000, F=16, W=160, H=112, seed=8, mu_s=math.log(0.05))
    perm = scenes.morton_order(sc.xyz)
    assert sorted(perm.tolist()) == list(range(3000)) and not torch.equal(perm, torch.arange(3000))
    sc2 = scenes.spatially_sorted(sc)
    assert torch.equal(sc2.xyz, sc.xyz[perm]) and torch.equal(sc2.seg_feature, sc.seg_feature[perm])
    assert torch.equal(sc2.labels3d, sc.labels3d[perm])
    # (activations are applied before the permutation here: vectorised CPU sigmoid / exp are not position-independent)
    inp2 = {k: (None if v is None else v[perm].contiguous()) for k, v in inp.items()}
    cam = cams[1]
    a_args, a = hip_forward(inp, cam, mode=MODE_EXACT)
    b_args, b = hip_forward(inp2, cam, mode=MODE_EXACT)
    assert a[0] == b[0]
    for k in (1, 2, 4):
        assert torch.equal(a[k], b[k])                      # colour, aux maps, feature map
    assert torch.equal(a[3][perm.cuda()], b[3])             # radii
    g = torch.Generator().manual_seed(1)
    dE = torch.randn(a[4].shape, generator=g).cuda()
    dC, dO = torch.zeros_like(a[1]), torch.zeros_like(a[2])
    ga = hip_backward(a_args, a, dC.cpu().numpy(), dO.cpu().numpy(), dE.cpu().numpy(), GRAD_EXTRA, MODE_EXACT)[8]
    gb = hip_backward(b_args, b, dC.cpu().numpy(), dO.cpu().numpy(), dE.cpu().numpy(), GRAD_EXTRA, MODE_EXACT)[8]
    assert torch.equal(ga[perm.cuda()], gb)


def test_forward_precomputed_colors_and_transmat():
    sc, cams, inp = small_scene(P=500, F=4, W=64, H=48, seed=13)
    cam = cams[1]
    st0 = oracle_forward(inp, cam)
    colors = np.random.RandomState(0).rand(500, 3).astype(np.float32)
    st = oracle_forward(inp, cam, colors_precomp=colors, shs=None)
    args, out = hip_forward(inp, cam, mode=MODE_EXACT, colors_precomp=colors)
    check_forward_exact(st, args, out)
    tm = st0["transMats"].copy()
    st2 = oracle_forward(inp, cam, transMat_precomp=tm, scales=None, rotations=None)
    args, out = hip_forward(inp, cam, mode=MODE_EXACT, transMat_precomp=tm)
    check_forward_exact(st2, args, out)


def test_forward_scale_modifier_and_sh_degrees():
    sc, cams, inp = small_scene(P=500, F=0, W=64, H=48, seed=14)
    for deg, mod in [(0, 1.0), (1, 0.7), (2, 1.3)]:
        st = oracle_forward(inp, cams[0], sh_degree=deg, scale_modifier=mod)
        args, out = hip_forward(inp, cams[0], sh_degree=deg, scale_modifier=mod, mode=MODE_EXACT)
        check_forward_exact(st, args, out)


def test_empty_and_fully_culled():
    sc, cams, inp = small_scene(P=50, F=3, W=32, H=32, seed=15)
    inp2 = dict(inp)
    inp2["means3D"] = inp["means3D"] + torch.tensor([100.0, 0.0, 0.0])     # everything off-screen / behind
    st = oracle_forward(inp2, cams[0], bg=(0.3, 0.2, 0.1))
    args, out = hip_forward(inp2, cams[0], bg=(0.3, 0.2, 0.1), mode=MODE_EXACT)
    check_forward_exact(st, args, out)
    # P = 0
    e = torch.empty(0, device="cuda")
    out = rz.rasterize_gaussians(torch.zeros(3, device="cuda"), torch.empty(0, 3, device="cuda"), e, e, e, e, 1.0, e, e, 0,
                                 cams[0].world_view_transform.cuda(), cams[0].full_proj_transform.cuda(), 0.5, 0.5, 32,
                                 32, torch.empty(0, 16, 3, device="cuda"), 3, cams[0].camera_center.cuda(), False, False)
    assert out[0] == 0 and float(out[1].abs().sum()) == 0.0


def _rand_grads(st, seed):
    g = np.random.RandomState(seed)
    dC = g.randn(*st["color"].shape).astype(np.float32)
    dO = g.randn(*st["others"].shape).astype(np.float32)
    dE = g.randn(*st["extra"].shape).astype(np.float32)
    return dC, dO, dE


def hip_backward(args, out, dC, dO, dE, mask, mode):
    R, color, others, radii, extra, geom, binning, img, grp, gidx = out
    return rz.rasterize_gaussians_backward(
        args["bg"], args["means3D"], radii, args["colors"], args["scales"], args["rotations"], args["extra_attrs"],
        args["scale_modifier"], args["transMat_precomp"], args["viewmatrix"], args["projmatrix"], args["tan_fovx"],
        args["tan_fovy"], torch.tensor(dC).cuda(), torch.tensor(dO).cuda(), torch.tensor(dE).cuda(), args["sh"],
        args["degree"], args["campos"], geom, R, binning, img, False, grad_mask=mask, mode=mode)


GRAD_NAMES = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dtransMat", "dL_dsh", "dL_dscales",
              "dL_drotations", "dL_dextra"]


@pytest.mark.parametrize("P,F,W,H,seed,bg", [
    (400, 6, 64, 48, 21, (0.0, 0.0, 0.0)),
    (1200, 32, 96, 64, 22, (0.3, 0.6, 0.1)),
    (900, 40, 80, 48, 23, (0.0, 0.0, 0.0)),
    (2500, 16, 112, 80, 24, (1.0, 1.0, 1.0)),
])
@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_backward_matches_oracle(P, F, W, H, seed, bg, mode):
    m = MODE_EXACT if mode == "exact" else MODE_FAST
    sc, cams, inp = small_scene(P=P, F=F, W=W, H=H, seed=seed, mu_s=math.log(0.05))
    cam = cams[2]
    st = oracle_forward(inp, cam, bg=bg)
    args, out = hip_forward(inp, cam, bg=bg, mode=m)
    dC, dO, dE = _rand_grads(st, seed)
    want = oracle.backward(st, dC, dO, dE)
    got = hip_backward(args, out, dC, dO, dE, GRAD_EXTRA | GRAD_GEOMETRY, m)
    for name, t in zip(GRAD_NAMES, got):
        assert_close(t.cpu().numpy().reshape(want[name].shape), want[name], 1e-3, f"{mode}:{name}")
    # features-only backward (train_semantic: only the feature parameter needs a gradient)
    got_e = hip_backward(args, out, dC, dO, dE, GRAD_EXTRA, m)
    want_e = oracle.backward(st, np.zeros_like(dC), np.zeros_like(dO), dE)
    assert_close(got_e[8].cpu().numpy(), want_e["dL_dextra"], 1e-3, f"{mode}:extra-only")
    assert got_e[0] is None and got_e[3] is None


def test_backward_is_run_to_run_deterministic():
    sc, cams, inp = small_scene(P=1500, F=16, W=96, H=64, seed=31, mu_s=math.log(0.06))
    st = oracle_forward(inp, cams[0])
    args, out = hip_forward(inp, cams[0], mode=MODE_EXACT)
    dC, dO, dE = _rand_grads(st, 1)
    a = hip_backward(args, out, dC, dO, dE, GRAD_EXTRA | GRAD_GEOMETRY, MODE_EXACT)
    b = hip_backward(args, out, dC, dO, dE, GRAD_EXTRA | GRAD_GEOMETRY, MODE_EXACT)
    for x, y in zip(a, b):
        assert torch.equal(x, y)        # no float atomics anywhere in the backward


def test_backward_precomputed_paths():
    sc, cams, inp = small_scene(P=500, F=4, W=64, H=48, seed=32)
    cam = cams[1]
    st0 = oracle_forward(inp, cam)
    colors = np.random.RandomState(0).rand(500, 3).astype(np.float32)
    tm = st0["transMats"].copy()
    st = oracle_forward(inp, cam, colors_precomp=colors, shs=None, transMat_precomp=tm, scales=None, rotations=None)
    args, out = hip_forward(inp, cam, mode=MODE_EXACT, colors_precomp=colors, transMat_precomp=tm)
    dC, dO, dE = _rand_grads(st, 2)
    want = oracle.backward(st, dC, dO, dE)
    got = hip_backward(args, out, dC, dO, dE, GRAD_EXTRA | GRAD_GEOMETRY, MODE_EXACT)
    for name in ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dtransMat", "dL_dextra"]:
        t = got[GRAD_NAMES.index(name)]
        assert_close(t.cpu().numpy().reshape(want[name].shape), want[name], 1e-3, name)


T_MARGIN = 1e-4        # relative distance of the oracle's own T from a T decision (the 1e-4 stop, the median's 0.5) below which a
                       # pixel may legitimately decide differently: FAST's T follows EXACT's to ~1e-6 per blended splat


def hip_forward_fast_counted(inp, cam, **kw):
    """FAST forward through the STATS build of the blend kernel: returns (args, out, counters).  counters[6] = (wave, splat)
    evaluations that took EXACT's instruction sequence, counters[7] = pairs OUTSIDE the guard bands whose decision differs from
    EXACT's (csrc/isr_fast_pair.hpp) - the band's bound, checked on the device for every pair: must be 0."""
    import ctypes
    from instascene_amd import _lib
    counters = torch.zeros(16, dtype=torch.int64, device="cuda")
    _lib.lib().isr_forward_set_counters(ctypes.c_void_p(counters.data_ptr()))
    args, out = hip_forward(inp, cam, mode=MODE_FAST, **kw)
    torch.cuda.synchronize()
    return args, out, [int(v) for v in counters.tolist()]


def fast_forward_by_cause(st, st_fma, out, dbg, counters=None, tol=1e-4):
    """The FAST forward against the oracle, gated by CAUSE, not by count:

    * (device) no pair outside the guard bands decides unlike EXACT (``counters[7] == 0``);
    * every pixel whose last / median contributor differs from the oracle's, or one of whose maps (colour, feature, depth, alpha,
      normal) is beyond ``tol`` of the map's max, must be EXPLAINED: the oracle's own T passed within ``T_MARGIN`` of a T decision
      (the T < 1e-4 stop, the median's T > 0.5 - T is a running product of FAST's own alphas: the two decisions FAST cannot replay
      with EXACT's arithmetic), or the oracle's
      second build (FMA contraction + libm expf: the latitude of the reference's own nvcc build) disagrees with its first on
      that pixel.  ``st`` must carry ``margins`` (``oracle_forward(..., margins=True)``), ``st_fma`` is the second build's state.

    Returns (explained, differing): boolean pixel maps [H, W]."""
    H, W = st["H"], st["W"]
    if counters is not None:
        assert counters[7] == 0, f"{counters[7]} pairs outside the guard bands decide unlike EXACT"
    differ = (dbg["n_contrib"] != st["n_contrib"]).any(axis=0).reshape(H, W)
    R, color, others, radii, extra = out[:5]
    for got, want in ((color, st["color"]), (extra, st["extra"]), (others[1], st["others"][1]), (others[0], st["others"][0]),
                      (others[2:5], st["others"][2:5])):
        if want.size == 0:
            continue
        g = got.cpu().numpy().reshape(-1, H, W)
        w = want.reshape(g.shape)
        differ |= (np.abs(g - w) > tol * np.abs(w).max()).any(axis=0)
    m = st["margins"]
    explained = (np.minimum(m[3], m[4]) < T_MARGIN).reshape(H, W)
    explained |= (st_fma["n_contrib"] != st["n_contrib"]).any(axis=0).reshape(H, W)
    explained |= (np.abs(st_fma["color"] - st["color"]) > tol * np.abs(st["color"]).max()).any(axis=0)
    bad = differ & ~explained
    assert not bad.any(), f"{int(bad.sum())} pixels differ from the oracle without a cause; first at (y, x) = {tuple(np.argwhere(bad)[0])}"
    return explained, differ


def rect_tiles(st, g):
    """Tiles of Gaussian g's rectangle (reference auxiliary.h:68-78)."""
    import oracle
    gx, gy = (st["W"] + 15) // 16, (st["H"] + 15) // 16
    x0, y0, x1, y1 = oracle.test_tile_rect(float(st["means2D"][g, 0]), float(st["means2D"][g, 1]), int(st["radii"][g]), gx, gy)
    return {(y, x) for y in range(y0, y1) for x in range(x0, x1)}


def rows_by_cause(name, got, want, st, explained, differ, row_dev=0.05):
    """Every row (Gaussian) of a FAST gradient beyond 1e-3 of the tensor's max must have an explained, differing pixel
    (fast_forward_by_cause) inside its tile rectangle, and stays within ``row_dev`` of the max.  Returns the rows."""
    w = np.asarray(want).reshape(st["P"], -1)
    g = np.asarray(got).reshape(w.shape)
    dev = np.abs(g - w).max(axis=1) / (np.abs(w).max() + 1e-30)
    out_rows = np.nonzero(dev > 1e-3)[0]
    if len(out_rows):
        ys, xs = np.nonzero(explained & differ)
        cause_tiles = {(int(y) // 16, int(x) // 16) for y, x in zip(ys, xs)}
        for r in out_rows:
            assert rect_tiles(st, int(r)) & cause_tiles, f"fast {name}: Gaussian {r} is off by {dev[r]:.3g} of the max without a cause"
            assert dev[r] <= row_dev, f"fast {name}: Gaussian {r} is off by {dev[r]:.3g} of the tensor's max"
    return [(name, int(r), float(dev[r])) for r in out_rows]


def _images_within_fast_tolerance(out, st, frac=1e-4, floor=2):
    """Within 1e-4 of the tensor's max on all but max(floor, frac * pixels) pixels; those within 1e-2.  (The count-based gate of
    rounds 1-3, kept for the opt-in variants - tight rectangles, precomputed inputs - whose forward is not gated by cause.)"""
    R, color, others, radii, extra = out[:5]
    for name, got, want in [("color", color, st["color"]), ("extra", extra, st["extra"]),
                            ("alpha", others[1], st["others"][1]), ("depth", others[0], st["others"][0]),
                            ("normal", others[2:5], st["others"][2:5])]:
        if want.size == 0:
            continue
        got = got.cpu().numpy().reshape(-1, st["H"], st["W"])
        want = want.reshape(got.shape)
        scale = np.abs(want).max()
        bad = (np.abs(got - want) > 1e-4 * scale).any(axis=0)          # a flipped decision shows in all of a pixel's channels
        assert bad.sum() <= max(floor, frac * bad.size), f"{name}: {bad.sum()} outlier pixels of {bad.size}"
        assert np.abs(got - want).max() <= 1e-2 * scale, f"{name}: {np.abs(got - want).max() / scale:.3g} of max"


@pytest.mark.parametrize("P,F,W,H,seed", [(1500, 16, 100, 70, 41), (3000, 32, 160, 112, 42), (2000, 0, 128, 96, 43),
                                          (2500, 40, 96, 80, 44), (1200, 72, 96, 64, 46), (1000, 128, 80, 64, 47)])
def test_fast_mode_keeps_the_reference_binning_bit_for_bit(P, F, W, H, seed):
    """The headline mode of bench.py: FAST arithmetic in the per-pixel loops (contracted FMAs, v_rcp / v_exp), the
    reference's tile rectangles.  radii, tiles_touched, point_list and ranges are bit-identical to the oracle's;
    images are within 1e-4 of the tensor's max except isolated pixels where a decision at the alpha = 1/255 or
    T = 1e-4 threshold flips (those stay within one skipped contribution)."""
    sc, cams, inp = small_scene(P=P, F=F, W=W, H=H, seed=seed, mu_s=math.log(0.05))
    for cam in cams[:2]:
        st = oracle_forward(inp, cam, bg=(0.1, 0.2, 0.3), margins=True)
        st2 = oracle_forward(inp, cam, bg=(0.1, 0.2, 0.3), fma=True)
        args, out, counters = hip_forward_fast_counted(inp, cam, bg=(0.1, 0.2, 0.3))
        dbg = check_binning_exact(st, out)
        # images within 1e-4 and the last / median contributors equal, except where the oracle itself sits on a T decision
        explained, differ = fast_forward_by_cause(st, st2, out, dbg, counters)
        assert differ.sum() <= 2
        # the instrumented build of the blend kernel (the one that checks every pair against EXACT) and the production build
        # produce the same bits
        args_p, out_p = hip_forward(inp, cam, bg=(0.1, 0.2, 0.3), mode=MODE_FAST)
        for k in (1, 2, 4):
            assert torch.equal(out_p[k], out[k]), k
        dbg_p = rz.debug_state(st["P"], st["W"], st["H"], out_p[0], out_p[5], out_p[6], out_p[7])
        np.testing.assert_array_equal(dbg_p["n_contrib"], dbg["n_contrib"])
        np.testing.assert_array_equal(dbg_p["final_T"], dbg["final_T"])


@pytest.mark.parametrize("P,F,W,H,seed", [(1500, 16, 100, 70, 41), (3000, 32, 160, 112, 42)])
def test_fast_tight_mode_forward_within_tolerance(P, F, W, H, seed):
    """The default FAST lists: a splat is binned only into the tiles it can reach (alpha >= 1/255 somewhere): every tile
    list is a subsequence, in the same order, of the reference's list; radii are untouched."""
    sc, cams, inp = small_scene(P=P, F=F, W=W, H=H, seed=seed, mu_s=math.log(0.05))
    st = oracle_forward(inp, cams[0], bg=(0.1, 0.2, 0.3))
    args, out = hip_forward(inp, cams[0], bg=(0.1, 0.2, 0.3), mode=MODE_FAST, tight=True)
    R, color, others, radii, extra = out[:5]
    assert 0 < R < st["R"]
    np.testing.assert_array_equal(radii.cpu().numpy(), st["radii"])
    dbg = rz.debug_state(P, W, H, R, out[5], out[6], out[7])
    ref_ranges = np.asarray(st["ranges"]).reshape(-1, 2)
    got_ranges = np.asarray(dbg["ranges"]).reshape(-1, 2)
    assert got_ranges.shape == ref_ranges.shape
    for (a0, a1), (b0, b1) in zip(got_ranges, ref_ranges):
        it = iter(st["point_list"][b0:b1])
        assert all(any(g == r for r in it) for g in dbg["point_list"][a0:a1])
    _images_within_fast_tolerance(out, st)


@pytest.mark.parametrize("P,F,W,H,seed,mu", [(1500, 16, 100, 70, 41, 0.05), (3000, 32, 160, 112, 42, 0.05),
                                             (6000, 0, 200, 150, 43, 0.02), (2500, 8, 130, 90, 44, 0.12)])
def test_tight_rectangles_change_no_output_bit(P, F, W, H, seed, mu):
    """FAST bins a splat only into the tiles its alpha >= 1/255 box reaches.  The per-block hit masks (k_pack_hits) apply the
    same box per 8x8 block, and a tile the box does not reach has none of its four blocks reached: the blend and backward
    kernels walk exactly the same (block, splat) pairs in the same order with the tight lists as with the reference's.  So
    nothing a caller can see differs - image, allmap channels 0-5, feature map, radii, the tracer pairs bit for bit, gradients
    to rounding (see below); only num_rendered (and the opaque state) shrink, and the distortion channel moves in its last bits."""
    sc, cams, inp = small_scene(P=P, F=F, W=W, H=H, seed=seed, mu_s=math.log(mu))
    bg = (0.1, 0.2, 0.3)
    a0, o0 = hip_forward(inp, cams[0], bg=bg, mode=MODE_FAST, tight=False, tracer=True)
    a1, o1 = hip_forward(inp, cams[0], bg=bg, mode=MODE_FAST, tight=True, tracer=True)
    assert 0 < o1[0] < o0[0]
    for k in (1, 3, 4):
        assert torch.equal(o0[k], o1[k]), k
    assert torch.equal(o0[2][:6], o1[2][:6])
    # (the distortion channel is evaluated relative to the depth of the tile's first list entry: same value, last bits move)
    assert float((o0[2][6] - o1[2][6]).abs().max()) <= 2e-6 * max(1.0, float(o0[2][6].abs().max()))
    n0, n1 = int(o0[9].item()) + 1, int(o1[9].item()) + 1
    assert n0 == n1
    key = lambda t: t[:, 0].to(torch.int64) * (1 << 32) + t[:, 1].to(torch.int64)
    assert torch.equal(torch.sort(key(o0[8][:n0])).values, torch.sort(key(o1[8][:n1])).values)
    st = dict(color=np.zeros((3, H, W), np.float32), others=np.zeros((7, H, W), np.float32), extra=np.zeros((F, H, W), np.float32))
    dC, dO, dE = _rand_grads(st, seed)
    dO[6] = 0.0                 # (its gradient reads the shifted moments: rounding-level differences)
    mask = (GRAD_EXTRA if F else 0) | GRAD_GEOMETRY
    g0 = hip_backward(a0, o0, dC, dO, dE, mask, MODE_FAST)
    g1 = hip_backward(a1, o1, dC, dO, dE, mask, MODE_FAST)
    # (bit-identical where the kernel chunks the list by hits - the splat-major geometry backward; the kernels that scan over
    # chunks of list POSITIONS associate their products differently on the shorter lists: rounding level)
    for name, x, y in zip(GRAD_NAMES, g0, g1):
        if x is not None and x.numel():
            assert float((x - y).abs().max()) <= 1e-6 * max(float(x.abs().max()), 1e-30), name


def test_homography_matches_the_reference_python(golden_dir):
    """K1 pinned by the reference's own Python (tests/golden/transmat.npz = the transMat_precomp its render() builds with
    pipe.compute_cov3D_python, gaussian_renderer/__init__.py:69-82): (i) the splat records k_preprocess writes hold the
    same Tu, Tv, Tw to fp32 rounding; (ii) a forward FROM the reference's matrices is bit-identical to the oracle's forward
    from them; (iii) render() with pipe.compute_cov3D_python takes that path end to end."""
    import os
    z = np.load(os.path.join(golden_dir, "transmat.npz"))
    c = np.load(os.path.join(golden_dir, "cameras.npz"))
    P = z["xyz"].shape[0]
    for i in range(4):
        Wc, Hc = (int(v) for v in c[f"wh{i}"])
        cam = scenes.Camera(Wc, Hc, float(c[f"fov{i}"][0]), float(c[f"fov{i}"][1]), torch.tensor(c[f"wvt{i}"]),
                            torch.tensor(c[f"proj{i}"]), torch.tensor(c[f"full{i}"]), torch.tensor(c[f"center{i}"]))
        for mod in ("1", "0.6", "1.7"):
            want = z[f"cam{i}_mod{mod}"]
            inp = dict(means3D=torch.tensor(z["xyz"]), opacities=torch.full((P, 1), 0.5),
                       scales=torch.exp(torch.tensor(z["log_scaling"])), rotations=torch.tensor(z["rotation_raw"]),
                       shs=None, extra=None)
            colors = np.full((P, 3), 0.5, np.float32)
            args, out = hip_forward(inp, cam, mode=MODE_EXACT, scale_modifier=float(mod), colors_precomp=colors,
                                    use_extra=False)
            radii = out[3].cpu().numpy()
            rec = rz.debug_state(P, Wc, Hc, out[0], out[5], out[6], out[7])["records"]
            seen = radii > 0
            if seen.sum():
                scale = np.abs(want[seen]).max(axis=1, keepdims=True)
                assert (np.abs(rec[seen, :9] - want[seen]) <= 2e-5 * scale).all(), (i, mod)
            st = oracle_forward(inp, cam, scale_modifier=1.0, transMat_precomp=want, scales=None, rotations=None,
                                colors_precomp=colors, shs=None)
            args2, out2 = hip_forward(inp, cam, mode=MODE_EXACT, transMat_precomp=want, colors_precomp=colors,
                                      use_extra=False)
            check_forward_exact(st, args2, out2)
    # (iii) through render(): the model's get_covariance -> _precomputed_transforms -> cov3D_precomp
    from instascene_amd.harness import PipelineParams, splat_to_world
    from instascene_amd.render import render

    class PC:
        active_sh_degree = 0
        get_xyz = torch.tensor(z["xyz"]).cuda()
        get_opacity = torch.full((P, 1), 0.5).cuda()
        get_features = torch.rand(P, 1, 3, generator=torch.Generator().manual_seed(1)).cuda()
        get_seg_feature = None

        def get_covariance(self, m=1):
            return splat_to_world(self.get_xyz, torch.exp(torch.tensor(z["log_scaling"])).cuda(), m,
                                  torch.tensor(z["rotation_raw"]).cuda())

    pipe = PipelineParams()
    pipe.compute_cov3D_python = True
    cam = scenes.Camera(*(int(v) for v in c["wh0"]), float(c["fov0"][0]), float(c["fov0"][1]), torch.tensor(c["wvt0"]),
                        torch.tensor(c["proj0"]), torch.tensor(c["full0"]), torch.tensor(c["center0"])).to("cuda")
    rz.set_mode("exact")
    pkg = render(cam, PC(), pipe, torch.zeros(3).cuda(), scaling_modifier=0.6)
    st = oracle.forward(z["xyz"], np.full((P, 1), 0.5, np.float32), c["wvt0"], c["full0"], c["center0"],
                        np.zeros(3, np.float32), cam.image_width, cam.image_height, math.tan(cam.FoVx / 2),
                        math.tan(cam.FoVy / 2), transMat_precomp=z["cam0_mod0.6"], shs=PC.get_features.cpu().numpy(),
                        sh_degree=0)
    assert_close(pkg["render"].cpu().numpy(), st["color"], 1e-4, "render() from precomputed transforms")


def test_mark_visible():
    sc, cams, inp = small_scene(P=2000, F=0, W=64, H=48, seed=51)
    cam = cams[0]
    want = oracle.mark_visible(inp["means3D"].numpy(), cam.world_view_transform.numpy(), cam.full_proj_transform.numpy())
    got = rz.mark_visible(inp["means3D"].cuda(), cam.world_view_transform.cuda(), cam.full_proj_transform.cuda())
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_autograd_module_matches_reference_interface():
    sc, cams, inp = small_scene(P=600, F=8, W=64, H=48, seed=61)
    cam = cams[0]
    settings = rz.GaussianRasterizationSettings(
        image_height=48, image_width=64, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.zeros(3, device="cuda"), scale_modifier=1.0, viewmatrix=cam.world_view_transform.cuda(),
        projmatrix=cam.full_proj_transform.cuda(), sh_degree=3, campos=cam.camera_center.cuda(), prefiltered=False,
        debug=False)
    r = rz.GaussianRasterizer(settings)
    with pytest.raises(Exception):
        r(means3D=inp["means3D"].cuda(), means2D=None, opacities=inp["opacities"].cuda())      # neither shs nor colours
    leaves = {k: v.cuda().requires_grad_(True) for k, v in inp.items()}
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    color, radii, allmap, extra, grp = r(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                                         shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"],
                                         extra_attrs=leaves["extra"])
    assert color.shape == (3, 48, 64) and allmap.shape == (7, 48, 64) and extra.shape == (8, 48, 64)
    assert radii.dtype == torch.int32 and grp.shape[1] == 2
    st = oracle_forward(inp, cam)
    g = torch.Generator().manual_seed(0)
    dC, dO, dE = torch.randn(3, 48, 64, generator=g), torch.randn(7, 48, 64, generator=g), torch.randn(8, 48, 64, generator=g)
    ((color * dC.cuda()).sum() + (allmap * dO.cuda()).sum() + (extra * dE.cuda()).sum()).backward()
    want = oracle.backward(st, dC.numpy(), dO.numpy(), dE.numpy())
    assert_close(leaves["means3D"].grad.cpu().numpy(), want["dL_dmeans3D"], 1e-3, "means3D.grad")
    assert_close(leaves["extra"].grad.cpu().numpy(), want["dL_dextra"], 1e-3, "extra.grad")
    assert_close(leaves["shs"].grad.cpu().numpy(), want["dL_dsh"], 1e-3, "shs.grad")
    assert_close(means2D.grad.cpu().numpy(), want["dL_dmeans2D"], 1e-3, "means2D.grad")
    assert_close(leaves["opacities"].grad.cpu().numpy(), want["dL_dopacity"], 1e-3, "opacity.grad")
    # frozen geometry (train_semantic): only the feature leaf gets a gradient
    feat = inp["extra"].cuda().requires_grad_(True)
    out = r(means3D=inp["means3D"].cuda(), means2D=torch.zeros(600, 3, device="cuda"), opacities=inp["opacities"].cuda(),
            shs=inp["shs"].cuda(), scales=inp["scales"].cuda(), rotations=inp["rotations"].cuda(), extra_attrs=feat)
    (out[3] * dE.cuda()).sum().backward()
    want_e = oracle.backward(st, np.zeros((3, 48, 64), np.float32), np.zeros((7, 48, 64), np.float32), dE.numpy())
    assert_close(feat.grad.cpu().numpy(), want_e["dL_dextra"], 1e-3, "feature-only grad")


def test_backward_with_sparse_upstream_gradient():
    """train_semantic samples a few thousand pixels: dL/dfeature is exactly zero elsewhere.  The backward skips
    zero-gradient pixels and culls splats against the live pixels' rectangle — results must not change."""
    sc, cams, inp = small_scene(P=2500, F=32, W=128, H=96, seed=81, mu_s=math.log(0.05))
    cam = cams[1]
    st = oracle_forward(inp, cam)
    for mode in (MODE_EXACT, MODE_FAST):
        args, out = hip_forward(inp, cam, mode=mode)
        rng = np.random.RandomState(5)
        dE = np.zeros_like(st["extra"])
        pix = rng.choice(128 * 96, 40, replace=False)
        dE.reshape(32, -1)[:, pix] = rng.randn(32, 40).astype(np.float32)
        dC, dO = np.zeros_like(st["color"]), np.zeros_like(st["others"])
        want = oracle.backward(st, dC, dO, dE)
        got = hip_backward(args, out, dC, dO, dE, GRAD_EXTRA, mode)
        assert_close(got[8].cpu().numpy(), want["dL_dextra"], 1e-3, "sparse extra-only")
        got = hip_backward(args, out, dC, dO, dE, GRAD_EXTRA | GRAD_GEOMETRY, mode)
        for name, t in zip(GRAD_NAMES, got):
            assert_close(t.cpu().numpy().reshape(want[name].shape), want[name], 1e-3, "sparse:" + name)
        # a single live pixel, and no live pixel at all
        dE1 = np.zeros_like(dE)
        dE1.reshape(32, -1)[:, pix[0]] = 1.0
        want1 = oracle.backward(st, dC, dO, dE1)
        got1 = hip_backward(args, out, dC, dO, dE1, GRAD_EXTRA, mode)
        assert_close(got1[8].cpu().numpy(), want1["dL_dextra"], 1e-3, "one pixel")
        got0 = hip_backward(args, out, dC, dO, np.zeros_like(dE), GRAD_EXTRA | GRAD_GEOMETRY, mode)
        assert all(float(t.abs().sum()) == 0.0 for t in got0)


@pytest.mark.parametrize("F", [8, 32, 48])
def test_backward_mixed_sparse_and_dense_tiles(F):
    """Feature-only backward: tiles with <= 32 live pixels go through the pixel-major (lane per splat) kernel, denser
    tiles through the MFMA kernel.  Cover both in one image (incl. exactly 32 / 33 live pixels in a tile), several
    feature chunk counts, and run-to-run determinism."""
    W, H = 112, 80
    sc, cams, inp = small_scene(P=2200, F=F, W=W, H=H, seed=33, mu_s=math.log(0.06))
    cam = cams[2]
    st = oracle_forward(inp, cam)
    rng = np.random.RandomState(11)
    dE = np.zeros_like(st["extra"]).reshape(F, H, W)
    dE[:, 0:16, 0:16] = rng.randn(F, 16, 16)                    # tile (0,0): fully dense
    ys, xs = np.divmod(rng.choice(256, 32, replace=False), 16)   # tile (1,0): exactly 32 live pixels
    dE[:, ys, 16 + xs] = rng.randn(F, 32)
    ys, xs = np.divmod(rng.choice(256, 33, replace=False), 16)   # tile (2,0): 33 -> dense path
    dE[:, ys, 32 + xs] = rng.randn(F, 33)
    pix = rng.choice(W * (H - 16), 60, replace=False) + 16 * W   # the rest: scattered samples
    dE.reshape(F, -1)[:, pix] = rng.randn(F, 60)
    dE[1:, 40, 50] = 0.0                                         # a pixel live through a single channel
    dE[0, 40, 50] = 0.7
    dE = dE.astype(np.float32).reshape(st["extra"].shape)
    dC, dO = np.zeros_like(st["color"]), np.zeros_like(st["others"])
    want = oracle.backward(st, dC, dO, dE)
    for mode in (MODE_EXACT, MODE_FAST):
        args, out = hip_forward(inp, cam, mode=mode)
        got = hip_backward(args, out, dC, dO, dE, GRAD_EXTRA, mode)[8]
        assert_close(got.cpu().numpy(), want["dL_dextra"], 1e-3, "mixed density F=%d" % F)
        again = hip_backward(args, out, dC, dO, dE, GRAD_EXTRA, mode)[8]
        assert torch.equal(got, again)


@pytest.mark.parametrize("W,H,F", [(90, 70, 8), (101, 67, 32), (130, 50, 20)])
def test_sparse_backward_on_ragged_image_sizes(W, H, F):
    """Widths that are not multiples of 4 (scalar path of the live-pixel scan), of the 16-pixel tile or of the 4-tile
    strip, and a feature width that is not a multiple of 32: sampled dL/dfeature incl. the last row / column."""
    sc, cams, inp = small_scene(P=1800, F=F, W=W, H=H, seed=57, mu_s=math.log(0.06))
    cam = cams[1]
    st = oracle_forward(inp, cam)
    rng = np.random.RandomState(W)
    dE = np.zeros_like(st["extra"]).reshape(F, -1)
    pix = np.concatenate([rng.choice(W * H, 50, replace=False), [W - 1, W * H - 1, W * (H - 1), 0]])
    dE[:, pix] = rng.randn(F, pix.size)
    dE = dE.astype(np.float32).reshape(st["extra"].shape)
    dC, dO = np.zeros_like(st["color"]), np.zeros_like(st["others"])
    want = oracle.backward(st, dC, dO, dE)
    for mode in (MODE_EXACT, MODE_FAST):
        args, out = hip_forward(inp, cam, mode=mode)
        got = hip_backward(args, out, dC, dO, dE, GRAD_EXTRA, mode)[8]
        assert_close(got.cpu().numpy(), want["dL_dextra"], 1e-3, "ragged %dx%d F=%d" % (W, H, F))


@pytest.mark.parametrize("F,W,H,nsamp", [(32, 112, 80, 300), (20, 101, 67, 90), (48, 64, 48, 2000)])
def test_sampled_feature_path_matches_dense_path(F, W, H, nsamp):
    """render(..., sample_pixels=): features gathered at sampled pixels and their gradient propagated without a dense
    [F,H,W] map (isr_sample_extra / isr_backward_sampled) == indexing the map and the ordinary backward.  Samples repeat,
    pile up in one tile (> 32: several groups) and include pixels nothing was blended into."""
    sc, cams, inp = small_scene(P=2000, F=F, W=W, H=H, seed=71, mu_s=math.log(0.06))
    cam = cams[1]
    st = oracle_forward(inp, cam)
    rng = np.random.RandomState(nsamp)
    pix = rng.randint(0, W * H, nsamp)
    pix[:70] = (rng.randint(0, 16, 70) * W + 16 + rng.randint(0, 16, 70))     # 70 samples in tile (1, 0), with repeats
    pix[70:75] = pix[0]
    g_rows = rng.randn(nsamp, F).astype(np.float32)
    dE = np.zeros((F, W * H), dtype=np.float32)
    np.add.at(dE.T, pix, g_rows)
    dC, dO = np.zeros_like(st["color"]), np.zeros_like(st["others"])
    want = oracle.backward(st, dC, dO, dE.reshape(st["extra"].shape))["dL_dextra"]
    pix_t = torch.tensor(pix, dtype=torch.int64).cuda()
    for mode in (MODE_EXACT, MODE_FAST):
        args, out = hip_forward(inp, cam, mode=mode)
        R, extra, geom, binning, img = out[0], out[4], out[5], out[6], out[7]
        feats = rz.sample_extra(extra, pix_t)
        assert torch.equal(feats, extra.reshape(F, -1)[:, pix_t].T)
        got = rz.rasterize_gaussians_backward_sampled(2000, F, W, H, R, pix_t, torch.tensor(g_rows).cuda(), None, geom, binning,
                                                      img, mode=mode)
        assert_close(got.cpu().numpy(), want, 1e-3, "sampled path F=%d" % F)
        again = rz.rasterize_gaussians_backward_sampled(2000, F, W, H, R, pix_t, torch.tensor(g_rows).cuda(), None, geom,
                                                        binning, img, mode=mode)
        if nsamp <= 300:          # deterministic whenever no tile holds more than 32 samples ... and in practice beyond
            assert torch.equal(got, again) or np.abs((got - again).cpu().numpy()).max() <= 1e-6 * np.abs(want).max()
        acc = rz.rasterize_gaussians_backward_sampled(2000, F, W, H, R, pix_t, torch.tensor(g_rows).cuda(), None, geom, binning,
                                                      img, accumulate_into=got.clone(), mode=mode)
        assert_close(acc.cpu().numpy(), 2.0 * want, 1e-3, "accumulate")


@pytest.mark.parametrize("F", [32, 20, 64])
def test_feature_rows_step_equals_the_three_passes(F):
    """isr_feature_rows_step (row reduction + chain rule through both normalisations + Adam + next normalisations in one
    pass) == isr_backward_sampled's reduction, then iso_rownorm2 backward, then iso_adam_rownorm2 — bit for bit; both the
    gradient-only form (multi-GPU) and the Adam form, with and without dL/dy and a dense dL/dz."""
    from instascene_amd.contrastive import FeatureAdam, _RowNorm2
    P, W, H, n = 2000, 112, 80, 700
    sc, cams, inp = small_scene(P=P, F=F, W=W, H=H, seed=23, mu_s=math.log(0.06))
    cam = cams[2]
    rng = np.random.RandomState(F)
    pix_t = torch.tensor(rng.randint(0, W * H, n), dtype=torch.int64).cuda()
    g_rows = torch.tensor(rng.randn(n, F).astype(np.float32)).cuda()
    x0 = torch.tensor(rng.randn(P, F).astype(np.float32)).cuda()
    x0[5] = 0.0                                                   # a zero row: the sub-gradient branch
    gy = torch.zeros(P, F).cuda()
    gy[rng.randint(0, P, 300)] = torch.tensor(rng.randn(300, F).astype(np.float32)).cuda()
    gzd = torch.tensor(rng.randn(P, F).astype(np.float32)).cuda() * 0.1
    for mode in (MODE_EXACT, MODE_FAST):
        args, out = hip_forward(inp, cam, mode=mode)
        R, geom, binning, img = out[0], out[5], out[6], out[7]
        ge = rz.rasterize_gaussians_backward_sampled(P, F, W, H, R, pix_t, g_rows, None, geom, binning, img, mode=mode)
        rows = rz.rasterize_gaussians_backward_sampled(P, F, W, H, R, pix_t, g_rows, None, geom, binning, img, mode=mode,
                                                       rows_only=True)
        for use_gy, use_gz, use_rows in [(True, False, True), (False, True, True), (True, True, False), (False, False, True)]:
            def make():
                p = torch.nn.Parameter(x0.clone())
                opt = FeatureAdam(p, lr=0.025, eps=1e-15)
                opt.exp_avg.copy_(torch.tensor(rng.randn(P, F).astype(np.float32)) * 0.01)
                opt.exp_avg_sq.copy_(torch.tensor(rng.rand(P, F).astype(np.float32)) * 1e-4)
                opt.step_count = 6
                return p, opt
            st = rng.get_state()
            p_a, opt_a = make()
            rng.set_state(st)
            p_b, opt_b = make()
            # reference: three passes
            y, z = _RowNorm2.apply(p_a, 1e-6, 1e-9)
            gz_total = (ge if use_rows else torch.zeros_like(ge)) + (gzd if use_gz else 0.0)
            torch.autograd.backward([y, z], [gy if use_gy else torch.zeros_like(gy), gz_total])
            want_grad = p_a.grad.clone()
            opt_a.step()
            # fused
            opt_b.leaf_mode = True
            for grad_only in (True, False):
                yl = opt_b.normalized_chain()
                zl = opt_b.leaves[1]
                if use_gy:
                    yl.grad = gy.clone()
                if use_gz:
                    zl.grad = gzd.clone()
                opt_b.step_rows(rows if use_rows else None, grad_only=grad_only)
                if grad_only:
                    assert torch.equal(p_b.grad, want_grad), (mode, use_gy, use_gz, use_rows)
                    p_b.grad = None
            assert torch.equal(p_b.data, p_a.data)
            assert torch.equal(opt_b.exp_avg, opt_a.exp_avg) and torch.equal(opt_b.exp_avg_sq, opt_a.exp_avg_sq)
            assert torch.equal(opt_b.normalized[1], opt_a.normalized[1]) and torch.equal(opt_b.normalized[2], opt_a.normalized[2])
            assert opt_b.step_count == opt_a.step_count == 7


@pytest.mark.parametrize("F", [64, 256, 8])
def test_feature_rows_step_wide_rows_and_row_ranges(F):
    """The one-pass tail with 16 / 64 / 2 lanes per row, and walked in row ranges (the multi-rank form): the ranges together
    equal the single call, for the gradient and for the Adam form."""
    from instascene_amd.contrastive import FeatureAdam
    from instascene_amd.dist_utils import row_ranges
    P = 1500
    rng = np.random.RandomState(F)
    x0 = torch.tensor(rng.randn(P, F).astype(np.float32)).cuda()
    gz = torch.tensor(rng.randn(P, F).astype(np.float32)).cuda()
    idx = torch.tensor(rng.randint(0, P, 400), dtype=torch.int64).cuda()
    vals = torch.tensor(rng.randn(400, F).astype(np.float32)).cuda()

    def fresh():
        p = torch.nn.Parameter(x0.clone())
        opt = FeatureAdam(p, lr=0.025, eps=1e-15)
        opt.leaf_mode = True
        y = opt.normalized_chain()
        opt.leaves[1].grad = gz.clone()
        return p, opt

    p_a, opt_a = fresh()
    tail = opt_a.begin_tail(None, (idx, vals))
    opt_a.tail_gradient(tail, 0, P)
    whole = p_a.grad.clone()
    p_b, opt_b = fresh()
    tail = opt_b.begin_tail(None, (idx, vals))
    for r0, r1 in row_ranges(P, 5):
        opt_b.tail_gradient(tail, r0, r1)
    assert torch.equal(p_b.grad, whole)
    # Adam: one call over all rows == begin_step / step_range ... / end_step on the same gradient
    p_c, opt_c = fresh()
    opt_c.tail_update(opt_c.begin_tail(None, (idx, vals)))
    opt_b.begin_step()
    for r0, r1 in row_ranges(P, 3):
        opt_b.step_range(r0, r1)
    opt_b.end_step()
    assert torch.equal(p_b.data, p_c.data) and torch.equal(opt_b.exp_avg_sq, opt_c.exp_avg_sq)
    assert torch.equal(opt_b.normalized[2], opt_c.normalized[2])


def test_sparse_row_gradient_equals_dense_index_put():
    """iso_rows_compact + isr_feature_rows_step(gy_slot, gy_merged) == the dense dL/dy that index_put_(accumulate=True)
    builds, bit for bit; rows drawn with replacement (pairs, a 5-fold repeat), out-of-range indices ignored."""
    from instascene_amd.contrastive import FeatureAdam
    P, F, n = 3000, 32, 1500
    rng = np.random.RandomState(3)
    idx = rng.randint(0, 400, n)                 # heavy repetition
    idx[[7, 300, 301, 900, 1499]] = 2999
    vals = torch.tensor(rng.randn(n, F).astype(np.float32)).cuda()
    idx_t = torch.tensor(idx, dtype=torch.int64).cuda()
    x0 = torch.tensor(rng.randn(P, F).astype(np.float32)).cuda()
    dense = torch.zeros(P, F).cuda().index_put_((idx_t,), vals, accumulate=True)
    outs = []
    for sparse in (False, True):
        p = torch.nn.Parameter(x0.clone())
        opt = FeatureAdam(p, lr=0.025, eps=1e-15)
        opt.leaf_mode = True
        yl = opt.normalized_chain()
        if not sparse:
            yl.grad = dense.clone()
        opt.step_rows(None, grad_only=True, row_grads=(idx_t, vals) if sparse else None)
        outs.append(p.grad.clone())
    assert torch.equal(outs[0], outs[1])
    assert outs[0].abs().max() > 0


@pytest.mark.parametrize("n,pool", [(16384, 6000), (32768, 20000), (32768, 3), (65536, 40000), (70000, 40000)])
def test_rows_compact_at_the_reference_default_batch(n, pool):
    """iso_rows_compact beyond round 4's 16 384 samples (the reference draws sample_batchsize = 32 768 Gaussians with replacement,
    arguments/__init__.py:103, train_semantic.py:183-190): one entry per distinct row, repeats summed in index order = the bits of
    index_put_(accumulate=True).  `pool = 3`: rows drawn ~11 000 times each (the repeat count needs the position field's width).
    Beyond 65 536 the torch fallback sums in atomic order: compared to 1e-6 there."""
    from instascene_amd.contrastive import compact_row_grads
    P, F = 50000, 16
    rng = np.random.RandomState(n + pool)
    idx = rng.randint(0, pool, n)
    idx[[5, n // 2, n - 1]] = P - 1
    idx[[9, n // 3]] = [-4, P + 7]                    # out of range: ignored
    vals = torch.tensor(rng.randn(n, F).astype(np.float32)).cuda()
    idx_t = torch.tensor(idx, dtype=torch.int64).cuda()
    ok = (idx_t >= 0) & (idx_t < P)
    dense = torch.zeros(P, F).cuda().index_put_((idx_t[ok],), vals[ok], accumulate=True)
    slot, merged = compact_row_grads(idx_t, vals, P)
    slot = slot.clone().long()
    got = torch.zeros(P, F).cuda()
    rows = (slot >= 0).nonzero().squeeze(1)
    got[rows] = merged[slot[rows]]
    assert rows.numel() == len(set(int(v) for v in idx if 0 <= v < P))
    if n <= 65536:
        assert torch.equal(got, dense)
        first = {}
        for i, v in enumerate(idx):
            first.setdefault(int(v), i)
        assert all(int(slot[r]) == first[int(r)] for r in rows[:200].tolist())
    else:
        assert (got - dense).abs().max().item() <= 1e-6 * max(1.0, dense.abs().max().item()) * 64
    # the table is the persistent one: hand it back clean for the next test
    from instascene_amd import contrastive as _c
    for t in _c._SLOT_TABLES.values():
        if t.slot.shape[0] == P:
            t.slot.fill_(-1); t.dirty = False


def test_sampled_feature_path_through_autograd():
    """GaussianRasterizer(..., sample_pixels=): sampled-only, sampled + dense map, and sampled with geometry gradients."""
    sc, cams, inp = small_scene(P=1500, F=16, W=96, H=64, seed=19, mu_s=math.log(0.06))
    cam = cams[0]
    rz.set_mode("exact")
    g = torch.Generator().manual_seed(4)
    pix = torch.randint(0, 96 * 64, (500,), generator=g).cuda()
    w_s = torch.randn(500, 16, generator=g).cuda()
    w_m = torch.randn(16, 64, 96, generator=g).cuda()

    def run(use_samples, use_map, geom_grad):
        t = {k: (v.cuda() if v is not None else None) for k, v in inp.items()}
        feat = t["extra"].clone().requires_grad_(True)
        xyz = t["means3D"].clone().requires_grad_(geom_grad)
        settings = rz.GaussianRasterizationSettings(64, 96, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3).cuda(), 1.0,
                                                    cam.world_view_transform.cuda(), cam.full_proj_transform.cuda(), 3,
                                                    cam.camera_center.cuda(), False, False)
        r = rz.GaussianRasterizer(settings)
        res = r(xyz, torch.zeros_like(xyz), t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"],
                extra_attrs=feat, sample_pixels=pix if use_samples else None)
        fmap = res[3]
        sampled = res[5] if use_samples else fmap.reshape(16, -1)[:, pix].T
        loss = (sampled * w_s).sum()
        if use_map:
            loss = loss + (fmap * w_m).sum()
        loss.backward()
        return feat.grad.clone(), (xyz.grad.clone() if geom_grad else None)

    for use_map, geom in [(False, False), (True, False), (False, True)]:
        ref_f, ref_x = run(False, use_map, geom)
        got_f, got_x = run(True, use_map, geom)
        assert_close(got_f.cpu().numpy(), ref_f.cpu().numpy(), 1e-4, "feature grad map=%s geom=%s" % (use_map, geom))
        if geom:
            assert_close(got_x.cpu().numpy(), ref_x.cpu().numpy(), 1e-4, "xyz grad")


def test_culling_survives_grazing_and_near_camera_splats():
    """Edge-on, huge and near-plane splats: the conservative cull box must never drop a contributing pair
    (EXACT mode stays bit-identical to the oracle, which evaluates every pair)."""
    g = torch.Generator().manual_seed(7)
    P = 1200
    sc, cams, inp = small_scene(P=P, F=8, W=96, H=80, seed=91)
    inp = dict(inp)
    # mix: very anisotropic (edge-on slivers), very large, and some close to the camera plane
    s = inp["scales"].clone()
    s[:300, 0] *= 40.0
    s[:300, 1] *= 0.02
    s[300:500] *= 25.0
    inp["scales"] = s
    xyz = inp["means3D"].clone()
    cam = cams[0]
    center = cam.camera_center
    fwd = -center / center.norm()
    xyz[500:700] = center + fwd * (0.21 + 0.3 * torch.rand(200, 1, generator=g)) + 0.2 * torch.randn(200, 3, generator=g)
    inp["means3D"] = xyz
    op = inp["opacities"].clone()
    op[::7] = 0.004            # barely above 1/255
    op[1::7] = 0.0039          # below: never contributes
    inp["opacities"] = op
    st = oracle_forward(inp, cam, bg=(0.5, 0.5, 0.5))
    args, out = hip_forward(inp, cam, bg=(0.5, 0.5, 0.5), mode=MODE_EXACT)
    check_forward_exact(st, args, out)
    dC, dO, dE = _rand_grads(st, 3)
    want = oracle.backward(st, dC, dO, dE)
    got = hip_backward(args, out, dC, dO, dE, GRAD_EXTRA | GRAD_GEOMETRY, MODE_EXACT)
    for name, t in zip(GRAD_NAMES, got):
        assert_close(t.cpu().numpy().reshape(want[name].shape), want[name], 2e-3, "grazing:" + name)


@pytest.mark.parametrize("F,W,H,mode", [(0, 256, 192, "fast"), (0, 512, 384, "exact"), (8, 512, 384, "fast")])
def test_splats_larger_than_the_view(F, W, H, mode):
    """A few surfels grown over the whole image (a train.py run produces them: backgrounds) touch every tile - 192 / 768
    rectangles here.  The geometry pass counts them, the key scatter places them and the per-Gaussian backward sums their
    hundreds of partial rows by the WORKGROUP (a tile / a row per thread), not in the lane that owns the splat: binning
    bit-identical to the oracle, forward exact / within the FAST tolerance, every gradient within 1e-3."""
    m = MODE_EXACT if mode == "exact" else MODE_FAST
    sc, cams, inp = small_scene(P=1500, F=F, W=W, H=H, seed=77, mu_s=math.log(0.04))
    cam = cams[1]
    st0 = oracle_forward(inp, cam)
    vis = np.nonzero(st0["radii"] > 0)[0]
    back = vis[np.argsort(-st0["depths"][vis])[:4]]   # the four farthest visible ones become backdrops
    scales = inp["scales"].clone()
    scales[torch.tensor(back.copy())] = torch.tensor([1.0, 2.0, 1.0, 2.0])[:, None]
    inp = dict(inp, scales=scales)
    st = oracle_forward(inp, cam)
    assert st["tiles_touched"].max() == ((W + 15) // 16) * ((H + 15) // 16) and (st["tiles_touched"] > 128).sum() >= 3
    args, out = hip_forward(inp, cam, mode=m)
    check_binning_exact(st, out)
    if mode == "exact":
        check_forward_exact(st, args, out)
    else:
        _images_within_fast_tolerance(out, st)
    dC, dO, dE = _rand_grads(st, 5)
    want = oracle.backward(st, dC, dO, dE)
    mask = GRAD_GEOMETRY | (GRAD_EXTRA if F else 0)
    got = hip_backward(args, out, dC, dO, dE, mask, m)
    for name, t in zip(GRAD_NAMES, got):
        if t is None or (name == "dL_dextra" and F == 0):
            continue
        assert_close(t.cpu().numpy().reshape(want[name].shape), want[name], 1e-3, f"{mode}:{name}")


def test_debug_mode_checks_every_launch_and_dumps_on_a_fault(tmp_path, monkeypatch):
    """``debug=True`` through the module: same results as without; with a fault injected at the library's third checked
    launch (``isr_set_debug``'s testing aid) the forward raises, names a kernel, and leaves snapshot_fw.dump; the backward
    likewise leaves snapshot_bw.dump."""
    sc, cams, inp = small_scene(P=500, F=8, W=64, H=48, seed=77)
    cam = cams[0]
    monkeypatch.chdir(tmp_path)

    def run(debug):
        st = rz.GaussianRasterizationSettings(image_height=48, image_width=64, tanfovx=math.tan(cam.FoVx / 2),
                                              tanfovy=math.tan(cam.FoVy / 2), bg=torch.zeros(3, device="cuda"), scale_modifier=1.0,
                                              viewmatrix=cam.world_view_transform.cuda(), projmatrix=cam.full_proj_transform.cuda(),
                                              sh_degree=3, campos=cam.camera_center.cuda(), prefiltered=False, debug=debug)
        leaves = {k: inp[k].cuda().requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "shs", "extra")}
        m2d = torch.zeros_like(leaves["means3D"], requires_grad=True)
        out = rz.GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=m2d, opacities=leaves["opacities"], shs=leaves["shs"],
                                        scales=leaves["scales"], rotations=leaves["rotations"], extra_attrs=leaves["extra"])
        return out, leaves

    (c0, r0, a0, e0, _), l0 = run(False)
    (c1, r1, a1, e1, _), l1 = run(True)
    assert torch.equal(c0, c1) and torch.equal(a0, a1) and torch.equal(e0, e1)
    (c0.sum() + e0.sum()).backward()
    (c1.sum() + e1.sum()).backward()
    for k in l0:
        assert torch.equal(l0[k].grad, l1[k].grad), k
    monkeypatch.setenv("ISR_DEBUG_FAULT_AFTER", "3")
    with pytest.raises(RuntimeError, match=r"\[debug\] kernel \w+ failed: injected fault"):
        run(True)
    assert (tmp_path / "snapshot_fw.dump").exists()
    snap = torch.load(tmp_path / "snapshot_fw.dump")
    assert torch.equal(snap[1], inp["means3D"])
    monkeypatch.delenv("ISR_DEBUG_FAULT_AFTER")
    (c2, _, _, e2, _), l2 = run(True)
    monkeypatch.setenv("ISR_DEBUG_FAULT_AFTER", "2")
    with pytest.raises(RuntimeError, match=r"\[debug\] kernel \w+ failed: injected fault"):
        (c2.sum() + e2.sum()).backward()
    assert (tmp_path / "snapshot_bw.dump").exists()


@pytest.mark.parametrize("F", [0, 4, 40])
def test_fast_mode_precomputed_paths_and_background(F):
    """FAST arithmetic on the optional-input paths (precomputed colours + precomputed transMat, non-zero background): F = 0
    takes the splat-major geometry backward, F = 4 / 40 the pixel-major one (one / two feature passes).  Binning identical
    to the oracle's, images within the FAST tolerance, gradients within 1e-3 of the tensor's maximum on all rows but a
    handful (threshold decisions, see tests/test_gpu_fuzz.py), and the colour adjoint identity with the background term."""
    sc, cams, inp = small_scene(P=900, F=max(F, 1), W=80, H=64, seed=52)
    if F == 0:
        inp = dict(inp, extra=None)
    cam = cams[2]
    bg = (0.3, 0.1, 0.6)
    st0 = oracle_forward(inp, cam, bg=bg)
    colors = np.random.RandomState(1).rand(900, 3).astype(np.float32)
    tm = st0["transMats"].copy()
    st = oracle_forward(inp, cam, bg=bg, colors_precomp=colors, shs=None, transMat_precomp=tm, scales=None, rotations=None)
    args, out = hip_forward(inp, cam, bg=bg, mode=MODE_FAST, colors_precomp=colors, transMat_precomp=tm)
    check_binning_exact(st, out)
    _images_within_fast_tolerance(out, st, frac=1e-3, floor=3)
    dC, dO, dE = _rand_grads(st, 4)
    want = oracle.backward(st, dC, dO, dE)
    mask = GRAD_GEOMETRY | (GRAD_EXTRA if F else 0)
    got = hip_backward(args, out, dC, dO, dE, mask, MODE_FAST)
    for name in ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dtransMat"] + (["dL_dextra"] if F else []):
        w = want[name].reshape(900, -1)
        g = got[GRAD_NAMES.index(name)].cpu().numpy().reshape(w.shape)
        dev = np.abs(g - w).max(axis=1) / (np.abs(w).max() + 1e-30)
        assert (dev > 1e-3).sum() <= 4 and dev.max() <= 0.05, (name, int((dev > 1e-3).sum()), float(dev.max()))
    # colour adjoint with a background: colour = sum w c + T bg, so <colour - T_final bg, dC> == <c, dL/dcolour>
    dbg = rz.debug_state(900, 80, 64, out[0], out[5], out[6], out[7])
    T_final = dbg["final_T"][0].reshape(64, 80)
    lin = out[1].cpu().numpy() - T_final[None] * np.asarray(bg, np.float32)[:, None, None]
    lhs = float((lin.astype(np.float64) * dC).sum())
    rhs = float((colors.astype(np.float64) * got[1].cpu().numpy()).sum())
    mag = float((np.abs(lin).astype(np.float64) * np.abs(dC)).sum())
    assert abs(lhs - rhs) <= 2e-5 * mag, (lhs, rhs, mag)


_TWO_KERNELS = r"""
import hashlib, math, sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from helpers import small_scene
from instascene_amd import rasterizer as rz
from instascene_amd._lib import MODE_FAST, MODE_EXACT
for mode in (MODE_FAST, MODE_EXACT):
  for P, F, W, H, seed in ((3000, 32, 160, 112, 42), (2000, 0, 128, 96, 43), (2500, 40, 97, 83, 44), (60000, 32, 320, 200, 45)):
      sc, cams, inp = small_scene(P=P, F=F, W=W, H=H, seed=seed, mu_s=math.log(0.05 if P < 10000 else 0.02))
      e = torch.empty(0, device="cuda")
      for cam in cams[:2]:
          out = rz.rasterize_gaussians(torch.tensor([0.1, 0.2, 0.3], device="cuda"), inp["means3D"].cuda(), e, inp["opacities"].cuda(),
                                       inp["scales"].cuda(), inp["rotations"].cuda(), 1.0, e, inp["extra"].cuda() if F else e, F,
                                       cam.world_view_transform.cuda(), cam.full_proj_transform.cuda(), math.tan(cam.FoVx / 2),
                                       math.tan(cam.FoVy / 2), H, W, inp["shs"].cuda(), 3, cam.camera_center.cuda(), False, False,
                                       mode=mode, tracer=True)
          h = hashlib.sha256()
          for t in (out[1], out[2], out[3], out[4]):          # colour, allmap, radii, feature map
              h.update(t.detach().cpu().numpy().tobytes())
          tr = out[8][: int(out[9]) + 1].detach().cpu().numpy()      # the tracer list: a SET of (gaussian, pixel) pairs
          h.update(tr[(tr[:, 0].astype("int64") * (W * H) + tr[:, 1]).argsort()].tobytes())
          print(h.hexdigest())
"""


def test_per_block_and_tile_wide_blend_kernels_produce_the_same_bits(tmp_path):
    """``k_render_fwd_fast_w`` / ``k_render_fwd_w`` (one wave per 8x8 block with its own hit list, hit masks from ``k_pack_hits``;
    the default) and ``k_render_fwd_fast`` / ``k_render_fwd<ExactMath>`` (the tile-wide kernels of rounds 1-3, ``ISR_FWD_WAVE=0``)
    share their pair arithmetic: every map is bit-identical and the tracer lists are equal as sets, in both arithmetic modes.  The switch is read once per
    process, hence two child processes."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "two_kernels.py"
    script.write_text(_TWO_KERNELS)
    outs = []
    for wave in ("1", "0"):
        env = dict(os.environ, ISR_FWD_WAVE=wave, ISR_MODE="fast")
        r = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if len(l) == 64])
    assert len(outs[0]) == 16 and outs[0] == outs[1]


_TWO_TILE_SCANS = r"""
import hashlib, math, sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from helpers import small_scene
from instascene_amd import rasterizer as rz
from instascene_amd._lib import MODE_FAST
for P, F, W, H, seed in ((40000, 8, 1296, 968, 51), (50000, 0, 1920, 1080, 52), (30000, 8, 1030, 1020, 53)):
    sc, cams, inp = small_scene(P=P, F=F, W=W, H=H, seed=seed, mu_s=math.log(0.03))
    e = torch.empty(0, device="cuda")
    cam = cams[1]
    out = rz.rasterize_gaussians(torch.tensor([0.1, 0.2, 0.3], device="cuda"), inp["means3D"].cuda(), e, inp["opacities"].cuda(),
                                 inp["scales"].cuda(), inp["rotations"].cuda(), 1.0, e, inp["extra"].cuda() if F else e, F,
                                 cam.world_view_transform.cuda(), cam.full_proj_transform.cuda(), math.tan(cam.FoVx / 2),
                                 math.tan(cam.FoVy / 2), H, W, inp["shs"].cuda(), 3, cam.camera_center.cuda(), False, False,
                                 mode=MODE_FAST, tracer=False)
    dbg = rz.debug_state(P, W, H, out[0], out[5], out[6], out[7])
    h = hashlib.sha256()
    for t in (out[1], out[2], out[3], out[4]):
        h.update(t.detach().cpu().numpy().tobytes())
    for k in ("tiles_touched", "point_list", "ranges", "n_contrib"):
        h.update(dbg[k].tobytes())
    print(h.hexdigest(), int(out[0]))
"""


def test_register_resident_tile_scan_equals_the_general_one(tmp_path):
    """Views of 4 096 .. 8 192 tiles take ``k_tile_scan_regs`` (a thread's eight tiles in registers: one round of loads);
    ``ISR_TILE_SCAN_REGS=0`` keeps the general kernel.  Same tile offsets, lists and images bit for bit at 81 x 61, 120 x 68 and
    65 x 64 tiles.  The switch is read once per process, hence two child processes."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "two_tile_scans.py"
    script.write_text(_TWO_TILE_SCANS)
    outs = []
    for regs in ("1", "0"):
        env = dict(os.environ, ISR_TILE_SCAN_REGS=regs, ISR_MODE="fast")
        r = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if len(l.split()) == 2 and len(l.split()[0]) == 64])
    assert len(outs[0]) == 3 and outs[0] == outs[1]
    assert all(int(l.split()[1]) > 0 for l in outs[0])
