"""Full-size checks (BASELINE.json config C3: 1.5 M Gaussians, 1920x1080, F = 32) through properties that need no oracle —
the CPU oracle takes ~5 s per C3 view and is used at small sizes elsewhere:

* the tile lists are a partition of the (Gaussian, tile) instances, sorted by (depth, id) inside every tile;
* the feature map is LINEAR in the features and the backward is its ADJOINT:  <render(E), G> == <E, backward(G)>  for a
  dense G (MFMA kernel) and for a G that lives on 16 384 sampled pixels (pixel-major kernel, sampled entry point);
* storing the Gaussians in Z-order changes no pixel whose splats have distinct depths (exact depth ties are broken by index);
* FAST binning is identical to EXACT binning (R, point_list, ranges), FAST images agree to 1e-4 on all but 1e-4 of the pixels;
* C3 and C1 at full size against the CPU oracle (forward bit-identical in EXACT; sampled feature backward / all gradients);
* 3-NN mean squared distance == brute force on a random subset of points;
* the batched contrastive losses: directional derivative by central differences.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

gpu = torch.cuda.is_available()
if gpu:
    from instascene_amd import rasterizer as rz, scenes
    from instascene_amd._lib import GRAD_EXTRA, MODE_EXACT, MODE_FAST
    from instascene_amd.contrastive import contrastive_loss_batch
    from instascene_amd.knn import distCUDA2

_CACHE = {}


def _c3():
    if "c3" not in _CACHE:
        scene, cams, cfg = scenes.config_scene("C3")
        inp = {k: (None if v is None else v.cuda()) for k, v in scenes.activated_inputs(scene).items()}
        _CACHE["c3"] = (scene, cams, cfg, inp)
    return _CACHE["c3"]


def _forward(inp, cam, cfg, mode, extra=None, tight=False):
    # (tight=False: the reference's tile rectangles, whose lists the oracle's positions - n_contrib - refer to; the default
    # FAST lists are subsequences of them: test_default_fast_lists_change_no_output_bit_at_full_size)
    e = torch.empty(0, device="cuda")
    ex = inp["extra"] if extra is None else extra
    args = (torch.zeros(3, device="cuda"), inp["means3D"], e, inp["opacities"], inp["scales"], inp["rotations"], 1.0, e, ex,
            ex.shape[1], cam.world_view_transform.cuda(), cam.full_proj_transform.cuda(), math.tan(cam.FoVx / 2),
            math.tan(cam.FoVy / 2), cfg["H"], cfg["W"], inp["shs"], 3, cam.camera_center.cuda(), False, False)
    return args, rz.rasterize_gaussians(*args, mode=mode, tracer=False, tight=tight)


def _backward_extra(args, out, dE, mode):
    R, color, others, radii, extra, geom, binning, img = out[:8]
    e = torch.empty(0, device="cuda")
    g = rz.rasterize_gaussians_backward(args[0], args[1], radii, e, args[4], args[5], args[8], 1.0, e, args[10], args[11],
                                        args[12], args[13], torch.zeros_like(color), torch.zeros_like(others), dE, args[16], 3,
                                        args[18], geom, R, binning, img, False, grad_mask=GRAD_EXTRA, mode=mode)
    return g[8]


def test_c3_tile_lists_are_a_sorted_partition():
    scene, cams, cfg, inp = _c3()
    P, W, H = cfg["P"], cfg["W"], cfg["H"]
    args, out = _forward(inp, cams[3], cfg, MODE_EXACT)
    R = out[0]
    dbg = rz.debug_state(P, W, H, R, out[5], out[6], out[7])
    tt, pl, rg = dbg["tiles_touched"].astype(np.int64), dbg["point_list"].astype(np.int64), dbg["ranges"].astype(np.int64)
    assert R == int(tt.sum()) and R > 4_000_000
    # ranges tile [0, R) without gaps or overlaps, in tile order
    lens = rg[:, 1] - rg[:, 0]
    ne = rg[lens > 0]                     # empty tiles carry (0, 0) like the reference's zero-initialised ranges
    assert (lens >= 0).all() and ne[0, 0] == 0 and ne[-1, 1] == R and (ne[1:, 0] == ne[:-1, 1]).all()
    assert int(lens.sum()) == R
    # every Gaussian appears exactly tiles_touched times
    assert np.array_equal(np.bincount(pl, minlength=P), tt)
    # inside a tile: ascending (depth bits, id); depths are positive floats, so their bit patterns order like the values
    depth_bits = dbg["records"][:, 18].view(np.uint32).astype(np.int64)
    key = (depth_bits[pl] << 32) | pl
    tile_of = np.repeat(np.arange(rg.shape[0]), lens)
    same_tile = tile_of[1:] == tile_of[:-1]
    assert (key[1:][same_tile] > key[:-1][same_tile]).all()
    # a Gaussian is listed at most once per tile (keys strictly increase, ids differ) and only in tiles of its rectangle
    assert int(dbg["n_contrib"][0].max()) <= int(lens.max())


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_c3_backward_is_the_adjoint_of_the_linear_feature_render(mode):
    scene, cams, cfg, inp = _c3()
    md = MODE_EXACT if mode == "exact" else MODE_FAST
    P, W, H, F = cfg["P"], cfg["W"], cfg["H"], cfg["F"]
    cam = cams[7]
    g = torch.Generator(device="cuda").manual_seed(11)
    E1 = torch.randn(P, F, device="cuda", generator=g)
    E2 = torch.randn(P, F, device="cuda", generator=g)
    a1, o1 = _forward(inp, cam, cfg, md, E1)
    a2, o2 = _forward(inp, cam, cfg, md, E2)
    a3, o3 = _forward(inp, cam, cfg, md, 0.5 * E1 - 2.0 * E2)
    # linearity in the features (the blend weights do not depend on them)
    lin = 0.5 * o1[4] - 2.0 * o2[4]
    assert float((o3[4] - lin).abs().max()) <= 2e-5 * float(lin.abs().max())
    assert torch.equal(o1[1], o2[1]) and torch.equal(o1[2], o2[2])          # colour / aux maps untouched by the features
    # adjoint, dense upstream gradient (MFMA backward)
    G = torch.randn(F, H, W, device="cuda", generator=g)
    lhs = float((o1[4].double() * G.double()).sum())
    dE = _backward_extra(a1, o1, G, md)
    rhs = float((E1.double() * dE.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), float((o1[4].double().abs() * G.double().abs()).sum()) * 1e-2)
    # adjoint, gradient on 16 384 sampled pixels: dense-map formulation (pixel-major kernel) and the sampled entry point
    pix = torch.randint(0, W * H, (16384,), device="cuda", generator=g)
    rows = torch.randn(16384, F, device="cuda", generator=g)
    Gs = torch.zeros(F, H * W, device="cuda")
    Gs.index_add_(1, pix, rows.t().contiguous())
    lhs_s = float((o1[4].reshape(F, -1).double() * Gs.double()).sum())
    dE_map = _backward_extra(a1, o1, Gs.reshape(F, H, W), md)
    R, geom, binning, img = o1[0], o1[5], o1[6], o1[7]
    dE_smp = rz.rasterize_gaussians_backward_sampled(P, F, W, H, R, pix, rows, None, geom, binning, img, mode=md)
    for name, d in (("dense map", dE_map), ("sampled", dE_smp)):
        rhs_s = float((E1.double() * d.double()).sum())
        assert abs(lhs_s - rhs_s) <= 2e-5 * float((o1[4].reshape(F, -1).double().abs() * Gs.double().abs()).sum()), name
    scale = float(dE_map.abs().max())
    assert float((dE_map - dE_smp).abs().max()) <= 1e-4 * scale
    # features gathered at the samples == indexing the map
    assert torch.equal(rz.sample_extra(o1[4], pix), o1[4].reshape(F, -1)[:, pix].t())


def test_c3_z_order_changes_no_pixel():
    scene, cams, cfg, inp = _c3()
    perm = scenes.morton_order(scene.xyz).cuda()
    inp2 = {k: (None if v is None else v[perm].contiguous()) for k, v in inp.items()}
    cam = cams[12]
    _, a = _forward(inp, cam, cfg, MODE_EXACT)
    _, b = _forward(inp2, cam, cfg, MODE_EXACT)
    assert a[0] == b[0]
    assert torch.equal(a[3][perm], b[3])
    # Bit-identical wherever the depth order of a pixel's splats is unambiguous.  Among 1.5 M Gaussians a few pairs in one
    # tile have EXACTLY equal view depths; the sort - like the reference's stable radix sort of (tile, depth) keys emitted in
    # index order - breaks such ties by the Gaussian's index, which the permutation changes, and front-to-back blending of
    # the two in the other order is a different (equally valid) result on the pixels they share: ~0.01 % of the image.
    for k in (1, 2, 4):
        changed = ((a[k] - b[k]).abs() > 0).any(dim=0)
        assert float(changed.float().mean()) < 1e-3, (k, float(changed.float().mean()))
    # and a second forward of the same inputs reproduces every bit
    _, c = _forward(inp2, cam, cfg, MODE_EXACT)
    assert torch.equal(b[1], c[1]) and torch.equal(b[4], c[4])


def test_c3_fast_mode_against_exact_mode():
    """FAST differs from EXACT only in the arithmetic of the per-pixel loops: the geometry pass and the binning are the same
    kernels, so radii, the instance count, point_list and ranges are identical at full size.  FAST's rho follows EXACT's to
    ~1e-6 (csrc/isr_fast_pair.hpp: EXACT's own roundings where they matter) and its alpha / near-plane / branch decisions are
    EXACT's by construction (guard bands), so the last and median contributors agree on at least 99.999 % of the 2 073 600
    pixels (measured on this view: 4 pixels differ, each one the T < 1e-4 stop or the median's T > 0.5 decided the other way by
    a T that differs in the 6th digit) and colour and feature agree to 1e-4 of the maximum on at least 99.999 % of them."""
    scene, cams, cfg, inp = _c3()
    P, W, H = cfg["P"], cfg["W"], cfg["H"]
    cam = cams[20]
    _, ex = _forward(inp, cam, cfg, MODE_EXACT)
    _, fa = _forward(inp, cam, cfg, MODE_FAST)
    assert torch.equal(ex[3], fa[3])                                         # radii
    assert fa[0] == ex[0]                                                    # R_fast == R_exact
    de = rz.debug_state(P, W, H, ex[0], ex[5], ex[6], ex[7])
    df = rz.debug_state(P, W, H, fa[0], fa[5], fa[6], fa[7])
    np.testing.assert_array_equal(df["tiles_touched"], de["tiles_touched"])
    np.testing.assert_array_equal(df["point_list"], de["point_list"])
    np.testing.assert_array_equal(df["ranges"], de["ranges"])
    same = (df["n_contrib"] == de["n_contrib"]).all(axis=0)
    assert same.mean() >= 0.99999, int((~same).sum())
    for k in (1, 4):
        ref = ex[k]
        bad = ((fa[k] - ref).abs() > 1e-4 * float(ref.abs().max())).any(dim=0)
        assert float(bad.float().mean()) <= 1e-5, (k, int(bad.sum()))


def test_c3_full_size_against_the_oracle():
    """BASELINE config 3 at FULL size against the CPU oracle (about 5 s per view on the test box's cores): the EXACT forward
    is bit-identical on every output and on all integer state; the sampled feature backward (the kernel the headline step
    lives on) is within 1e-3 of the tensor's maximum on every row in EXACT mode, with the 99.9th percentile of the per-row
    relative error within 1e-2; FAST mode is gated by cause (test_gpu_rasterizer.fast_forward_by_cause / rows_by_cause): no pair
    outside the guard bands decides unlike EXACT (checked on the device for all ~4.5e8 evaluated pairs), every differing pixel and
    every gradient row beyond 1e-3 traces to a pixel where the oracle's T sits within 1e-4 of a T decision or where the oracle's
    two builds disagree."""
    import oracle
    import test_gpu_rasterizer as TR
    from helpers import oracle_forward, assert_rows_close
    scene, cams, cfg, inp = _c3()
    P, W, H, F = cfg["P"], cfg["W"], cfg["H"], cfg["F"]
    cam = cams[9]
    cpu = {k: (None if v is None else v.cpu()) for k, v in inp.items()}
    st = oracle_forward(cpu, cam, margins=True)
    st2 = oracle_forward(cpu, cam, fma=True)
    a, o = _forward(inp, cam, cfg, MODE_EXACT)
    assert o[0] == st["R"]
    np.testing.assert_array_equal(o[3].cpu().numpy(), st["radii"])
    dbg = rz.debug_state(P, W, H, o[0], o[5], o[6], o[7])
    for k in ("tiles_touched", "point_list", "ranges", "n_contrib", "final_T"):
        np.testing.assert_array_equal(dbg[k], st[k], err_msg=k)
    np.testing.assert_array_equal(o[1].cpu().numpy(), st["color"])
    np.testing.assert_array_equal(o[2].cpu().numpy(), st["others"])
    np.testing.assert_array_equal(o[4].cpu().numpy(), st["extra"])
    g = torch.Generator(device="cuda").manual_seed(31)
    pix = torch.randint(0, W * H, (16384,), device="cuda", generator=g)
    rows = torch.randn(16384, F, device="cuda", generator=g)
    Gs = torch.zeros(F, H * W, device="cuda")
    Gs.index_add_(1, pix, rows.t().contiguous())
    want = oracle.backward(st, np.zeros((3, H, W), np.float32), np.zeros((7, H, W), np.float32),
                           Gs.reshape(F, H, W).cpu().numpy())["dL_dextra"]
    scale = np.abs(want).max()
    explained = differ = None
    for md in (MODE_EXACT, MODE_FAST):
        if md == MODE_FAST:
            a, o, counters = TR.hip_forward_fast_counted(cpu, cam)
            df = rz.debug_state(P, W, H, o[0], o[5], o[6], o[7])
            explained, differ = TR.fast_forward_by_cause(st, st2, o, df, counters)
            assert differ.sum() <= 2e-5 * W * H, int(differ.sum())
        got = rz.rasterize_gaussians_backward_sampled(P, F, W, H, o[0], pix, rows, None, o[5], o[6], o[7], mode=md).cpu().numpy()
        dev = np.abs(got - want).max(axis=1) / scale
        if md == MODE_EXACT:
            assert dev.max() <= 1e-3, float(dev.max())
            assert_rows_close(got, want, "C3 exact dL_dextra")
        else:
            out_rows = TR.rows_by_cause("dL_dextra", got, want, st, explained, differ)
            assert len(out_rows) <= 1e-5 * P, out_rows


def test_c1_plumbing_config_against_the_oracle():
    """BASELINE config 1 at its stated size (50 k Gaussians, 256 x 256, RGB only): EXACT forward bit-identical, all
    gradients within 1e-3; FAST gated by cause like the fuzz sweep (tests/test_gpu_fuzz.py)."""
    import oracle
    import test_gpu_rasterizer as TR
    from helpers import oracle_forward, assert_rows_close
    scene, cams, cfg = scenes.config_scene("C1")
    assert (cfg["P"], cfg["W"], cfg["H"], cfg["F"]) == (50_000, 256, 256, 0)
    inp = scenes.activated_inputs(scene)
    cam = cams[2]
    st = oracle_forward(inp, cam, bg=(0.3, 0.2, 0.1), tracer=True, margins=True)
    st2 = oracle_forward(inp, cam, bg=(0.3, 0.2, 0.1), fma=True)
    args, out = TR.hip_forward(inp, cam, bg=(0.3, 0.2, 0.1), mode=MODE_EXACT, tracer=True)
    TR.check_forward_exact(st, args, out, tracer=True)
    rng = np.random.RandomState(1)
    dC = rng.randn(3, 256, 256).astype(np.float32)
    dO = rng.randn(7, 256, 256).astype(np.float32)
    dE = np.zeros((0, 256, 256), np.float32)
    want = oracle.backward(st, dC, dO, None)
    explained = differ = None
    for md in (MODE_EXACT, MODE_FAST):
        if md == MODE_EXACT:
            args, out = TR.hip_forward(inp, cam, bg=(0.3, 0.2, 0.1), mode=md)
        else:
            args, out, counters = TR.hip_forward_fast_counted(inp, cam, bg=(0.3, 0.2, 0.1))
            explained, differ = TR.fast_forward_by_cause(st, st2, out, TR.check_binning_exact(st, out), counters)
        got = TR.hip_backward(args, out, dC, dO, dE, TR.GRAD_GEOMETRY, md)
        for name, t in zip(TR.GRAD_NAMES, got):
            if t is None or name not in want or want[name].size == 0:
                continue
            w = want[name].reshape(cfg["P"], -1)
            gg = t.cpu().numpy().reshape(w.shape)
            dev = np.abs(gg - w).max(axis=1) / (np.abs(w).max() + 1e-30)
            if md == MODE_EXACT:
                assert dev.max() <= 1e-3, (name, float(dev.max()))
                assert_rows_close(gg, w, f"C1 exact {name}")
            else:
                TR.rows_by_cause(name, gg, w, st, explained, differ)


def test_c3_three_nearest_neighbours_against_brute_force():
    scene, cams, cfg, inp = _c3()
    pts = inp["means3D"]
    got = distCUDA2(pts)
    g = torch.Generator(device="cuda").manual_seed(5)
    probe = torch.randint(0, pts.shape[0], (512,), device="cuda", generator=g)
    d2 = torch.cdist(pts[probe].double(), pts.double()) ** 2               # [512, P]
    d2[torch.arange(512, device="cuda"), probe] = float("inf")
    want = d2.topk(3, dim=1, largest=False).values.mean(dim=1)
    assert float((got[probe].double() - want).abs().max()) <= 1e-5 * float(want.max())
    assert float(got.min()) > 0.0 and bool(torch.isfinite(got).all())


def test_full_batch_contrastive_losses_directional_derivative():
    g = torch.Generator(device="cuda").manual_seed(3)
    N, F, K = 8192, 32, 65
    feats = [torch.nn.functional.normalize(torch.randn(N, F, device="cuda", generator=g), dim=1) for _ in range(3)]
    labels = [torch.randint(0, K, (N,), device="cuda", generator=g) for _ in range(3)]
    pre = torch.nn.functional.normalize(torch.randn(K, F, device="cuda", generator=g), dim=1)
    predefs, w = [None, pre, pre], [0.5, 1.0, 2.5]
    leaves = [f.clone().requires_grad_(True) for f in feats]
    total, parts = contrastive_loss_batch(leaves, labels, predefs, w, num_labels=K)
    total.backward()
    assert float(parts[:3].sum()) == pytest.approx(float(total.detach()), rel=1e-6)
    dirs = [torch.randn(N, F, device="cuda", generator=g) for _ in range(3)]
    analytic = sum(float((l.grad.double() * d.double()).sum()) for l, d in zip(leaves, dirs))
    eps = 2e-3
    with torch.no_grad():
        up = contrastive_loss_batch([f + eps * d for f, d in zip(feats, dirs)], labels, predefs, w, num_labels=K)[0]
        dn = contrastive_loss_batch([f - eps * d for f, d in zip(feats, dirs)], labels, predefs, w, num_labels=K)[0]
    numeric = (float(up) - float(dn)) / (2 * eps)
    assert abs(numeric - analytic) <= 2e-2 * abs(analytic) + 1e-3 * abs(float(total.detach()))


def test_c3_trainer_formulations_agree_bit_for_bit():
    """The bench configuration itself (C3, FAST, async binning, Z-order): four steps with every trainer-level extension on
    == the same steps with the plain formulation (dense gradient map, separate reduction / normalisation / Adam kernels,
    one launch sequence per loss, no prefetch) - identical losses and parameters; and the multi-rank form of the tail."""
    from instascene_amd.harness import SegTrainer
    scene, cams, cfg, inp = _c3()
    rz.set_mode("fast")
    rz.set_tracer(False)
    rz.set_async_binning(True)
    try:
        outs = []
        for variant in ("all", "plain_tail", "split_tail"):
            kw = {}
            if variant == "plain_tail":
                kw = dict(fused_tail=False, batched_losses=False, prefetch_geometry=False)
            tr = SegTrainer(scene, cams[:4], device="cuda", sample_batchsize=8192, use_class_feat=True, seed=1, **kw)
            tr.split_tail = variant == "split_tail"
            losses = [float(tr.step(it)) for it in range(4)]
            outs.append((losses, tr.model._seg_feature.detach().clone()))
            del tr
            torch.cuda.empty_cache()
        for other in outs[1:]:
            assert outs[0][0] == other[0]
            assert torch.equal(outs[0][1], other[1])
    finally:
        rz.set_async_binning(False)
        rz.set_mode("exact")
        rz.set_tracer(True)


def test_c5_full_size_against_the_oracle():
    """BASELINE config 5 in its 1-GPU form (5 M Gaussians, 1296 x 968, F = 64: ~15 M tile instances, lists of ~3 100 per tile -
    the dense-scene paths of the binning: k_tile_sort_big, two-chunk feature passes) at FULL size against the CPU oracle: the
    EXACT forward bit-identical on every output and all integer state; the sampled feature backward within 1e-3 of the tensor's
    maximum on every row in EXACT mode; FAST gated by cause like C3."""
    import oracle
    import test_gpu_rasterizer as TR
    from helpers import oracle_forward, assert_rows_close
    scene, cams, cfg = scenes.config_scene("C5")
    P, W, H, F = cfg["P"], cfg["W"], cfg["H"], cfg["F"]
    assert (P, W, H, F) == (5_000_000, 1296, 968, 64)
    cpu = scenes.activated_inputs(scene)
    inp = {k: (None if v is None else v.cuda()) for k, v in cpu.items()}
    cam = cams[4]
    st = oracle_forward(cpu, cam, margins=True)
    st2 = oracle_forward(cpu, cam, fma=True)
    a, o = _forward(inp, cam, cfg, MODE_EXACT)
    assert o[0] == st["R"] and st["R"] > 12_000_000
    np.testing.assert_array_equal(o[3].cpu().numpy(), st["radii"])
    dbg = rz.debug_state(P, W, H, o[0], o[5], o[6], o[7])
    for k in ("tiles_touched", "point_list", "ranges", "n_contrib", "final_T"):
        np.testing.assert_array_equal(dbg[k], st[k], err_msg=k)
    np.testing.assert_array_equal(o[1].cpu().numpy(), st["color"])
    np.testing.assert_array_equal(o[2].cpu().numpy(), st["others"])
    np.testing.assert_array_equal(o[4].cpu().numpy(), st["extra"])
    del dbg
    g = torch.Generator(device="cuda").manual_seed(37)
    pix = torch.randint(0, W * H, (16384,), device="cuda", generator=g)
    rows = torch.randn(16384, F, device="cuda", generator=g)
    Gs = torch.zeros(F, H * W, device="cuda")
    Gs.index_add_(1, pix, rows.t().contiguous())
    want = oracle.backward(st, np.zeros((3, H, W), np.float32), np.zeros((7, H, W), np.float32),
                           Gs.reshape(F, H, W).cpu().numpy())["dL_dextra"]
    del Gs
    scale = np.abs(want).max()
    got = rz.rasterize_gaussians_backward_sampled(P, F, W, H, o[0], pix, rows, None, o[5], o[6], o[7], mode=MODE_EXACT).cpu().numpy()
    dev = np.abs(got - want).max(axis=1) / scale
    assert dev.max() <= 1e-3, float(dev.max())
    assert_rows_close(got, want, "C5 exact dL_dextra")
    del a, o
    a, o, counters = TR.hip_forward_fast_counted(cpu, cam)
    df = rz.debug_state(P, W, H, o[0], o[5], o[6], o[7])
    explained, differ = TR.fast_forward_by_cause(st, st2, o, df, counters)
    assert differ.sum() <= 2e-5 * W * H, int(differ.sum())
    got = rz.rasterize_gaussians_backward_sampled(P, F, W, H, o[0], pix, rows, None, o[5], o[6], o[7], mode=MODE_FAST).cpu().numpy()
    out_rows = TR.rows_by_cause("dL_dextra", got, want, st, explained, differ)
    assert len(out_rows) <= 1e-5 * P, out_rows


def test_c5_two_feature_passes_adjoint_and_trainer():
    """Config C5 (5 M Gaussians, 1296x968, F = 64): the feature channels take two 32-channel passes through every blend
    kernel.  Adjoint identity of the sampled backward, and the trainer's fused tail against the plain formulation."""
    from instascene_amd.harness import SegTrainer
    scene, cams, cfg = scenes.config_scene("C5")
    P, W, H, F = cfg["P"], cfg["W"], cfg["H"], cfg["F"]
    assert F == 64 and P == 5_000_000
    inp = {k: (None if v is None else v.cuda()) for k, v in scenes.activated_inputs(scene).items()}
    g = torch.Generator(device="cuda").manual_seed(2)
    E = torch.randn(P, F, device="cuda", generator=g)
    a, o = _forward(inp, cams[5], cfg, MODE_FAST, E)
    pix = torch.randint(0, W * H, (16384,), device="cuda", generator=g)
    rows = torch.randn(16384, F, device="cuda", generator=g)
    sampled = rz.sample_extra(o[4], pix)
    lhs = float((sampled.double() * rows.double()).sum())
    dE = rz.rasterize_gaussians_backward_sampled(P, F, W, H, o[0], pix, rows, None, o[5], o[6], o[7], mode=MODE_FAST)
    rhs = float((E.double() * dE.double()).sum())
    assert abs(lhs - rhs) <= 2e-5 * float((sampled.double().abs() * rows.double().abs()).sum())
    del inp, E, a, o, dE
    torch.cuda.empty_cache()
    rz.set_mode("fast")
    rz.set_tracer(False)
    rz.set_async_binning(True)
    try:
        outs = []
        for plain in (False, True):
            kw = dict(fused_tail=False, batched_losses=False, prefetch_geometry=False) if plain else {}
            tr = SegTrainer(scene, cams[:3], device="cuda", sample_batchsize=8192, use_class_feat=True, seed=1, **kw)
            losses = [float(tr.step(it)) for it in range(3)]
            outs.append((losses, tr.model._seg_feature.detach().clone()))
            del tr
            torch.cuda.empty_cache()
        assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])
    finally:
        rz.set_async_binning(False)
        rz.set_mode("exact")
        rz.set_tracer(True)


# ---- BASELINE config 2 at full size (300 k Gaussians, 779 x 519, RGB + depth + normal): the dense backward WITH geometry
def _c2():
    if "c2" not in _CACHE:
        scene, cams, cfg = scenes.config_scene("C2")
        inp = {k: (None if v is None else v.cuda()) for k, v in scenes.activated_inputs(scene).items()}
        _CACHE["c2"] = (scene, cams, cfg, inp)
    return _CACHE["c2"]


def _render_c2(inp, cam, cfg, mode, colors=None, over=None, tight=False):
    from instascene_amd._lib import GRAD_GEOMETRY  # noqa: F401
    e = torch.empty(0, device="cuda")
    v = dict(inp)
    if over:
        v.update(over)
    args = (torch.tensor([0.2, 0.1, 0.3], device="cuda"), v["means3D"], e if colors is None else colors, v["opacities"],
            v["scales"], v["rotations"], 1.0, e, e, 0, cam.world_view_transform.cuda(), cam.full_proj_transform.cuda(),
            math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), cfg["H"], cfg["W"], v["shs"] if colors is None else e, 3,
            cam.camera_center.cuda(), False, False)
    return args, rz.rasterize_gaussians(*args, mode=mode, tracer=False, tight=tight)


def _backward_c2(args, out, dC, dO, mode):
    from instascene_amd._lib import GRAD_GEOMETRY
    R, color, others, radii, extra, geom, binning, img = out[:8]
    e = torch.empty(0, device="cuda")
    return rz.rasterize_gaussians_backward(args[0], args[1], radii, args[2], args[4], args[5], e, 1.0, e, args[10], args[11],
                                           args[12], args[13], dC, dO, e, args[16], 3, args[18], geom, R, binning, img, False,
                                           grad_mask=GRAD_GEOMETRY, mode=mode)


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_c2_geometry_backward_adjoint_and_oracle_parity_at_full_size(mode):
    """Full-size C2 through the dense backward with GRAD_GEOMETRY (k_render_bwd<GEOM> + k_preprocess_bwd):
    (i) the colour image is LINEAR in per-Gaussian colours and dL/dcolors is its adjoint: <render(c) - render(0), G> ==
    <c, dL/dcolors(G)>; (ii) forward and all gradients against the CPU oracle on the full-size view."""
    scene, cams, cfg, inp = _c2()
    md = MODE_EXACT if mode == "exact" else MODE_FAST
    P, W, H = cfg["P"], cfg["W"], cfg["H"]
    cam = cams[5]
    g = torch.Generator(device="cuda").manual_seed(21)
    # (i) colours
    c1 = torch.rand(P, 3, device="cuda", generator=g)
    a1, o1 = _render_c2(inp, cam, cfg, md, colors=c1)
    a0, o0 = _render_c2(inp, cam, cfg, md, colors=torch.zeros(P, 3, device="cuda"))
    G = torch.randn(3, H, W, device="cuda", generator=g)
    lhs = float(((o1[1] - o0[1]).double() * G.double()).sum())
    grads = _backward_c2(a1, o1, G, torch.zeros(7, H, W, device="cuda"), md)
    rhs = float((c1.double() * grads[1].double()).sum())
    assert abs(lhs - rhs) <= 2e-5 * float(((o1[1] - o0[1]).double().abs() * G.double().abs()).sum())
    # (ii) against the CPU oracle at FULL C2 size (0.7 s per view on the test box's cores): the forward in EXACT mode is
    # bit-identical, every gradient of the dense geometry backward is within 1e-3 of the tensor's maximum; in FAST mode
    # the binning is bit-identical and at most 1e-4 of the rows may sit outside 1e-3 (threshold decisions, see
    # tests/test_gpu_fuzz.py).  (A finite-difference check is useless at this size: the render is a sum over 4 * 10^5
    # pixels of terms that jump at the alpha = 1/255 and T = 1e-4 thresholds, and the jumps crossed inside any usable step
    # are as large as the derivative itself - measured: central differences at two step sizes disagree by 20-100 %.)
    import oracle
    import test_gpu_rasterizer as TR
    from helpers import oracle_forward
    cpu = {k: (None if v is None else v.cpu()) for k, v in inp.items()}
    st = oracle_forward(cpu, cam, bg=(0.2, 0.1, 0.3))
    a, o = _render_c2(inp, cam, cfg, md)
    if mode == "exact":
        assert o[0] == st["R"]
        np.testing.assert_array_equal(o[1].cpu().numpy(), st["color"])
        np.testing.assert_array_equal(o[2].cpu().numpy(), st["others"])
        np.testing.assert_array_equal(o[3].cpu().numpy(), st["radii"])
    dbg = rz.debug_state(P, W, H, o[0], o[5], o[6], o[7])
    np.testing.assert_array_equal(dbg["point_list"], st["point_list"])
    np.testing.assert_array_equal(dbg["ranges"], st["ranges"])
    rng = np.random.RandomState(3)
    dC = rng.randn(3, H, W).astype(np.float32)
    dO = rng.randn(7, H, W).astype(np.float32)
    want = oracle.backward(st, dC, dO, None)
    got = _backward_c2(a, o, torch.tensor(dC).cuda(), torch.tensor(dO).cuda(), md)
    for name, t in zip(TR.GRAD_NAMES, got):
        if t is None or name not in want or want[name].size == 0:
            continue
        w = want[name].reshape(P, -1)
        dev = np.abs(t.cpu().numpy().reshape(w.shape) - w).max(axis=1) / (np.abs(w).max() + 1e-30)
        if mode == "exact":
            assert dev.max() <= 1e-3, (name, float(dev.max()))
            from helpers import assert_rows_close
            assert_rows_close(t.cpu().numpy().reshape(w.shape), w, f"C2 exact {name}")
        else:
            assert (dev > 1e-3).sum() <= 1e-4 * P and dev.max() <= 0.05, (name, int((dev > 1e-3).sum()), float(dev.max()))


def _same_outputs(o0, o1):
    """Everything a caller sees of two forwards, bit for bit - except the distortion channel (allmap[6]), whose shifted-moment
    evaluation is anchored at the depth of the tile's first list entry: equal to ~1e-6 of its range."""
    assert torch.equal(o0[1], o1[1]) and torch.equal(o0[3], o1[3]) and torch.equal(o0[4], o1[4])
    assert torch.equal(o0[2][:6], o1[2][:6])
    d0, d1 = o0[2][6], o1[2][6]
    assert float((d0 - d1).abs().max()) <= 2e-6 * max(1.0, float(d0.abs().max()))


def test_default_fast_lists_change_no_output_bit_at_full_size():
    """The default FAST mode bins a splat only where its alpha >= 1/255 box reaches (rasterizer._CONFIG["tight_rects"]); the
    kernels then walk the same (block, splat) pairs as on the reference's rectangles (tests/test_gpu_rasterizer.py::
    test_tight_rectangles_change_no_output_bit).  Here at full size: the C3 view (forward + the sampled feature backward of the
    headline step) and the C2 view (forward + the dense geometry backward of train.py) - outputs bit for bit, the dense
    gradients too, the sampled ones to rounding (1e-6 of the maximum), with 8-19 % fewer tile instances."""
    scene, cams, cfg, inp = _c3()
    P, W, H, F = cfg["P"], cfg["W"], cfg["H"], cfg["F"]
    cam = cams[11]
    a0, o0 = _forward(inp, cam, cfg, MODE_FAST, tight=False)
    a1, o1 = _forward(inp, cam, cfg, MODE_FAST, tight=True)
    assert 0.5 * o0[0] < o1[0] < 0.95 * o0[0], (o0[0], o1[0])
    _same_outputs(o0, o1)
    g = torch.Generator(device="cuda").manual_seed(5)
    pix = torch.randint(0, W * H, (16384,), device="cuda", generator=g)
    rows = torch.randn(16384, F, device="cuda", generator=g)
    e = torch.empty(0, device="cuda")
    g0 = rz.rasterize_gaussians_backward_sampled(P, F, W, H, o0[0], pix, rows, e, o0[5], o0[6], o0[7], mode=MODE_FAST)
    g1 = rz.rasterize_gaussians_backward_sampled(P, F, W, H, o1[0], pix, rows, e, o1[5], o1[6], o1[7], mode=MODE_FAST)
    # (the sampled backward scans the transmittance over chunks of 64 list POSITIONS: with shorter lists the chunk boundaries
    # fall elsewhere and the products associate differently - measured 3.6e-7 against a maximum of 4.6; the splat-major dense
    # backward below chunks by hits and is bit-identical)
    assert float((g0 - g1).abs().max()) <= 1e-6 * float(g0.abs().max())
    del a0, o0, a1, o1, g0, g1
    scene, cams, cfg, inp = _c2()
    W, H = cfg["W"], cfg["H"]
    cam = cams[7]
    a0, o0 = _render_c2(inp, cam, cfg, MODE_FAST, tight=False)
    a1, o1 = _render_c2(inp, cam, cfg, MODE_FAST, tight=True)
    assert o1[0] < o0[0]
    _same_outputs(o0, o1)
    dC = torch.randn(3, H, W, device="cuda", generator=g)
    dO = torch.randn(7, H, W, device="cuda", generator=g)
    dO[6] = 0.0                 # (the distortion channel's gradient reads the shifted moments: rounding-level differences)
    for x, y in zip(_backward_c2(a0, o0, dC, dO, MODE_FAST), _backward_c2(a1, o1, dC, dO, MODE_FAST)):
        if x is not None and x.numel():
            assert torch.equal(x, y)
    del a0, o0, a1, o1
    # C5 (lists of ~3 000 per tile: the radix-sorted buckets, two feature passes of 32 channels): forward outputs
    scene, cams, cfg = scenes.config_scene("C5")
    inp = {k: (None if v is None else v.cuda()) for k, v in scenes.activated_inputs(scene).items()}
    _, o0 = _forward(inp, cams[4], cfg, MODE_FAST, tight=False)
    _, o1 = _forward(inp, cams[4], cfg, MODE_FAST, tight=True)
    assert o1[0] < o0[0]
    _same_outputs(o0, o1)
