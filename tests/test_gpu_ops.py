"""GPU parity for the companion ops: HIP contrastive loss vs the reference's own outputs (golden
fixtures) and vs the torch oracle on seeded inputs; HIP 3-NN vs brute force; render() dict."""
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import pytest
import torch

import oracle
from oracle import torch_ops
from helpers import assert_close, small_scene

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from instascene_amd.contrastive import contrastive_loss
    from instascene_amd.knn import distCUDA2
    from instascene_amd.render import render
    from instascene_amd import rasterizer as rz


@pytest.mark.parametrize("tag", ["computed", "predef", "negative", "f32dim", "minpix"])
def test_contrastive_matches_reference_golden(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, "contrastive_loss.npz"))
    f = torch.tensor(z[f"{tag}_features"]).cuda().requires_grad_(True)
    lab = torch.tensor(z[f"{tag}_labels"]).cuda()
    kw = {}
    if tag == "minpix":
        kw["min_pixnum"] = int(z["minpix_min_pixnum"])
    else:
        pre = z[f"{tag}_predef"]
        kw["predef_u_list"] = torch.tensor(pre).cuda() if pre.size else None
        kw["consider_negative"] = bool(z[f"{tag}_consider_negative"])
    loss = contrastive_loss(f, lab, **kw)
    loss.backward()
    want = float(z[f"{tag}_loss"])
    assert abs(float(loss.detach()) - want) <= 1e-4 * abs(want)
    assert_close(f.grad.cpu().numpy(), z[f"{tag}_grad"], 1e-3, tag + " grad")


@pytest.mark.parametrize("N,F,K,predef", [(8192, 32, 64, False), (8192, 32, 64, True), (5000, 16, 37, False),
                                          (3000, 64, 130, True), (777, 6, 5, False)])
def test_contrastive_matches_oracle_on_seeded_inputs(N, F, K, predef):
    g = torch.Generator().manual_seed(N + F + K)
    feats = torch.randn(N, F, generator=g)
    labels = torch.randint(0, K + 1, (N,), generator=g)
    pre = torch.nn.functional.normalize(torch.randn(K + 1, F, generator=g), dim=1) if predef else None
    a = feats.clone().requires_grad_(True)
    want = torch_ops.contrastive_loss(a, labels, predef_u=pre)
    (want * 0.37).backward()
    b = feats.cuda().requires_grad_(True)
    got = contrastive_loss(b, labels.cuda(), predef_u_list=None if pre is None else pre.cuda())
    (got * 0.37).backward()
    assert abs(float(got.detach()) - float(want.detach())) <= 1e-4 * abs(float(want.detach()))
    assert_close(b.grad.cpu().numpy(), a.grad.numpy(), 1e-3, "grad")
    # deterministic: no float atomics
    c = feats.cuda().requires_grad_(True)
    got2 = contrastive_loss(c, labels.cuda(), predef_u_list=None if pre is None else pre.cuda())
    (got2 * 0.37).backward()
    assert torch.equal(got2.detach(), got.detach()) and torch.equal(c.grad, b.grad)


@pytest.mark.parametrize("P,kind", [(5000, "uniform"), (4097, "clustered"), (3000, "planar"), (10, "uniform"),
                                    (3, "uniform"), (2000, "duplicates")])
def test_dist2_3nn_matches_brute_force(P, kind):
    g = np.random.RandomState(P)
    if kind == "uniform":
        pts = g.rand(P, 3).astype(np.float32) * 3 - 1.5
    elif kind == "clustered":
        c = g.randn(8, 3) * 2
        pts = (c[g.randint(0, 8, P)] + g.randn(P, 3) * 0.02).astype(np.float32)
        pts[:5] += 50.0                               # far outliers: ring expansion
    elif kind == "planar":
        pts = np.concatenate([g.rand(P, 2), np.zeros((P, 1))], 1).astype(np.float32)
    else:
        pts = g.rand(P // 2, 3).astype(np.float32)
        pts = np.concatenate([pts, pts], 0)           # exact duplicates -> zero distances
    want = oracle.dist2_3nn(pts)
    got = distCUDA2(torch.tensor(pts).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, want)          # same fp32 distance expression, exact neighbours


def test_dist2_regular_grid_closed_form():
    g = np.stack(np.meshgrid(np.arange(6), np.arange(6), np.arange(6), indexing="ij"), -1).reshape(-1, 3)
    d = distCUDA2(torch.tensor(g, dtype=torch.float32).cuda()).cpu().numpy().reshape(6, 6, 6)
    assert d[2, 3, 2] == 1.0 and d[0, 0, 0] == 1.0


class _PC:
    def __init__(self, inp, active_sh_degree=3):
        self._i = {k: (v.cuda() if v is not None else None) for k, v in inp.items()}
        self.active_sh_degree = active_sh_degree
    get_xyz = property(lambda s: s._i["means3D"])
    get_opacity = property(lambda s: s._i["opacities"])
    get_scaling = property(lambda s: s._i["scales"])
    get_rotation = property(lambda s: s._i["rotations"])
    get_features = property(lambda s: s._i["shs"])
    get_seg_feature = property(lambda s: s._i["extra"])


class _Pipe:
    compute_cov3D_python = False
    convert_SHs_python = False
    depth_ratio = 1.0
    debug = False


def test_render_returns_reference_dict_and_matches_oracle():
    from helpers import oracle_forward
    sc, cams, inp = small_scene(P=700, F=8, W=64, H=48, seed=71)
    import copy
    cam = cams[0]
    pc = _PC(inp)
    camg = copy.deepcopy(cams[0]).to("cuda")
    rz.set_mode("exact")
    out = render(camg, pc, _Pipe(), torch.zeros(3, device="cuda"))
    keys = {"render", "viewspace_points", "visibility_filter", "radii", "seg_feature", "gau_related_pixels",
            "rend_alpha", "rend_normal", "rend_dist", "surf_depth", "surf_normal", "rend_depth", "rend_median_depth"}
    assert set(out.keys()) == keys
    # render() re-normalises the feature with +1e-9 (reference :61-62) before rasterising
    # (normalised with the same HIP row-normalise op render() uses, so the oracle sees identical bits)
    from instascene_amd.contrastive import row_normalize
    feat = row_normalize(inp["extra"].cuda(), 1e-9).cpu()
    st = oracle_forward(dict(inp, extra=feat), cams[0])
    np.testing.assert_array_equal(out["render"].cpu().numpy(), st["color"])
    np.testing.assert_array_equal(out["seg_feature"].cpu().numpy(), st["extra"])
    np.testing.assert_array_equal(out["radii"].cpu().numpy(), st["radii"])
    want = torch_ops.render_post(torch.tensor(st["others"]), cam.world_view_transform.cpu(), cam.full_proj_transform.cpu(),
                                 64, 48, 1.0)
    for k, v in want.items():
        assert_close(out[k].cpu().numpy(), v.numpy(), 1e-4, k)


def test_inplace_operations_on_render_outputs():
    """The reference's rasterizer returns fresh tensors; user code may clamp or scale them in place.  With the arena on, the
    outputs must behave the same (round 4 leased them as views created inside the autograd.Function: `image.clamp_()` raised)."""
    import copy
    from instascene_amd import arena
    sc, cams, inp = small_scene(P=3000, F=8, W=640, H=480, seed=72)       # maps above arena.MIN_BYTES: leased, not torch.empty
    pc = _PC({k: (v.cuda().requires_grad_(True) if (v is not None and v.is_floating_point()) else v) for k, v in inp.items()})
    camg = copy.deepcopy(cams[0]).to("cuda")
    rz.set_mode("fast")
    before = arena.STATS["leases"]
    out = render(camg, pc, _Pipe(), torch.zeros(3, device="cuda"))
    assert not arena.ENABLED or arena.STATS["leases"] > before
    img, feat = out["render"], out["seg_feature"]
    assert img.requires_grad and img._base is None
    ref = img.detach().clone()
    img.clamp_(0.0, 0.5)
    feat[1:2].mul_(2.0)
    out["rend_alpha"].add_(1.0)
    assert torch.equal(img.detach(), ref.clamp(0.0, 0.5))
    (img.sum() + feat.sum()).backward()
    assert pc._i["extra"].grad is not None and torch.isfinite(pc._i["extra"].grad).all()
    rz.set_mode("exact")


@pytest.mark.parametrize("N,F,eps", [(5000, 32, 1e-6), (777, 16, 1e-9), (300, 6, 1e-6), (100, 64, 1e-9)])
def test_row_normalize_matches_torch(N, F, eps):
    from instascene_amd.contrastive import row_normalize
    g = torch.Generator().manual_seed(N)
    x = torch.randn(N, F, generator=g)
    x[3] = 0.0                                   # zero row: subgradient 0, no NaN
    dy = torch.randn(N, F, generator=g)
    a = x.clone().requires_grad_(True)
    ya = a / (a.norm(dim=-1, keepdim=True) + eps)
    (ya * dy).sum().backward()
    b = x.cuda().requires_grad_(True)
    yb = row_normalize(b, eps)
    (yb * dy.cuda()).sum().backward()
    assert_close(yb.detach().cpu().numpy(), ya.detach().numpy(), 1e-6, "rownorm fwd")
    assert_close(b.grad.cpu().numpy(), a.grad.numpy(), 1e-5, "rownorm bwd")


@pytest.mark.parametrize("N,F,use", [(5000, 32, "yz"), (777, 16, "z"), (300, 64, "y"), (129, 40, "yz")])
def test_row_normalize_chain_matches_two_passes(N, F, use):
    """The getter + render() normalisations fused: forward bit-identical to two row_normalize passes, the second
    row_normalize call is a lookup, backward equal to torch autograd of the chain (either upstream gradient absent)."""
    from instascene_amd.contrastive import row_normalize, row_normalize_chain
    g = torch.Generator().manual_seed(N + 1)
    x = torch.randn(N, F, generator=g) * 3.0
    x[5] = 0.0
    gy, gz = torch.randn(N, F, generator=g), torch.randn(N, F, generator=g)
    a = x.clone().double().requires_grad_(True)
    ya = a / (a.norm(dim=-1, keepdim=True) + 1e-6)
    za = ya / (ya.norm(dim=-1, keepdim=True) + 1e-9)
    loss = 0.0
    if "y" in use: loss = loss + (ya * gy.double()).sum()
    if "z" in use: loss = loss + (za * gz.double()).sum()
    loss.backward()
    b = x.cuda().requires_grad_(True)
    yb = row_normalize_chain(b, 1e-6, 1e-9)
    zb = row_normalize(yb, 1e-9)                       # served from the chain
    assert zb.grad_fn is yb.grad_fn
    y2 = row_normalize(x.cuda(), 1e-6)
    assert torch.equal(yb.detach(), y2) and torch.equal(zb.detach(), row_normalize(y2, 1e-9))
    lb = 0.0
    if "y" in use: lb = lb + (yb * gy.cuda()).sum()
    if "z" in use: lb = lb + (zb * gz.cuda()).sum()
    lb.backward()
    assert_close(b.grad.cpu().numpy(), a.grad.float().numpy(), 2e-5, "rownorm chain bwd (%s)" % use)
    assert row_normalize(yb, 1e-6) is not zb           # another eps is a fresh normalisation


@pytest.mark.parametrize("N,F,K", [(4096, 32, 65), (1000, 16, 13), (700, 48, 130)])
def test_contrastive_batch_equals_separate_losses(N, F, K):
    """iso_contrastive_forward/backward_batch: three losses (cluster-mean prototypes, predefined, predefined with unlabeled
    samples) in one sequence of launches == the three single calls, weighted and added in order — bit for bit, values
    and gradients (the small-K MFMA path and the general path)."""
    from instascene_amd.contrastive import contrastive_loss_batch
    g = torch.Generator().manual_seed(N + K)
    feats = [torch.randn(N, F, generator=g).cuda() for _ in range(3)]
    labels = [torch.randint(1, K, (N,), generator=g).cuda() for _ in range(3)]
    labels[2][::7] = 0
    pre = torch.nn.functional.normalize(torch.randn(K, F, generator=g), dim=1).cuda()
    predefs = [None, pre, pre]
    w = [5e-7, 1e-6, 2.5e-6]
    singles = [f.clone().requires_grad_(True) for f in feats]
    want = None
    for f, l, u, wi in zip(singles, labels, predefs, w):
        term = contrastive_loss(f, l, predef_u_list=u, num_labels=K) * wi
        want = term if want is None else want + term
    want.backward()
    batched = [f.clone().requires_grad_(True) for f in feats]
    got, parts = contrastive_loss_batch(batched, labels, predefs, w, num_labels=K)
    got.backward()
    assert float(got.detach()) == float(want.detach())
    assert float(parts[3]) == float(got.detach())
    for a, b in zip(batched, singles):
        assert torch.equal(a.grad, b.grad)
    # the first two problems stacked in one [2N,F] input (what a trainer's sampled rows of one render are): same value, and the
    # stacked input's gradient is the two gradients one after the other - without autograd concatenating anything
    stacked = torch.cat([feats[0], feats[1]]).requires_grad_(True)
    third = feats[2].clone().requires_grad_(True)
    got2, parts2 = contrastive_loss_batch([stacked, third], labels, predefs, w, num_labels=K, stacked=2)
    got2.backward()
    assert float(got2.detach()) == float(want.detach()) and torch.equal(parts2, parts)
    assert torch.equal(stacked.grad, torch.cat([singles[0].grad, singles[1].grad])) and torch.equal(third.grad, singles[2].grad)
    with pytest.raises(ValueError):
        contrastive_loss_batch([stacked[:-1], third], labels, predefs, w, num_labels=K, stacked=2)
    # a batch of one is the plain loss
    one = feats[1].clone().requires_grad_(True)
    l1, _ = contrastive_loss_batch([one], [labels[1]], [pre], [1.0], num_labels=K)
    assert float(l1.detach()) == float(contrastive_loss(feats[1], labels[1], predef_u_list=pre, num_labels=K))


def test_contrastive_with_label_bound_and_dropped_samples():
    """num_labels bound larger than the labels present, unlabeled (0) samples, min_pixnum dropping small clusters."""
    g = torch.Generator().manual_seed(5)
    N, F = 4000, 32
    feats = torch.randn(N, F, generator=g)
    labels = torch.randint(0, 40, (N,), generator=g)
    labels[labels == 7] = 0
    labels[:3] = 55                                # a tiny cluster (3 samples) dropped by min_pixnum
    a = feats.clone().requires_grad_(True)
    want = torch_ops.contrastive_loss(a, labels, min_pixnum=5)
    want.backward()
    b = feats.cuda().requires_grad_(True)
    got = contrastive_loss(b, labels.cuda(), min_pixnum=5, num_labels=200)
    got.backward()
    assert abs(float(got.detach()) - float(want.detach())) <= 1e-4 * abs(float(want.detach()))
    assert_close(b.grad.cpu().numpy(), a.grad.numpy(), 1e-3, "grad")
    assert float(b.grad[labels.cuda() == 0].abs().sum()) == 0.0


def test_async_binning_and_lazy_tracer_slice():
    """Binning workspace sized from THIS view's verified instance count of an earlier forward (no blocking read of R)
    gives identical results; a view seen for the first time is sized exactly; when the estimate turns out too small an
    eval render is redone with the exact size, a training forward raises BinningOverflow at its backward - and in both
    cases the estimate is corrected and async binning stays on.  The tracer list is sliced only when accessed."""
    import copy
    from helpers import oracle_forward
    sc, cams, inp = small_scene(P=900, F=8, W=64, H=48, seed=77)
    pc = _PC(inp)
    rz.set_mode("exact")
    rz.set_tracer(True)
    rz._R_ESTIMATE.clear(); rz._PENDING.clear(); rz._OVERFLOWED.clear()
    try:
        rz.set_async_binning(True)
        camg = [copy.deepcopy(cams[k]).to("cuda") for k in range(2)]       # long-lived cameras, like a trainer's
        bg = torch.zeros(3, device="cuda")
        outs = []
        with torch.no_grad():
            for k in range(4):                   # first visit of each view sizes exactly, revisits run async
                outs.append(render(camg[k % 2], pc, _Pipe(), bg))
        assert len(rz._R_ESTIMATE) == 2          # one verified count per view
        feat = __import__("instascene_amd.contrastive", fromlist=["row_normalize"]).row_normalize(inp["extra"].cuda(), 1e-9).cpu()
        st0 = oracle_forward(dict(inp, extra=feat), cams[0], tracer=True)
        np.testing.assert_array_equal(outs[2]["render"].cpu().numpy(), st0["color"])
        np.testing.assert_array_equal(outs[2]["seg_feature"].cpu().numpy(), st0["extra"])
        grp = outs[2]["gau_related_pixels"]       # lazily sliced here
        assert grp.shape[0] == len(st0["tracer"])
        assert {(int(a), int(b)) for a, b in grp.cpu().numpy()} == {(int(a), int(b)) for a, b in st0["tracer"]}
        # force an overflow: pretend view 0 had almost no instances, then render it with much larger splats
        key0 = next(k for k in rz._R_ESTIMATE if k[-2:] == rz._view_id(camg[0].world_view_transform, camg[0].full_proj_transform))
        slack, rz._ASYNC_SLACK = rz._ASYNC_SLACK, 0
        big = {k: (v.clone() if v is not None else None) for k, v in inp.items()}
        big["scales"] = big["scales"] * 30.0        # far more tile instances than 1 * 1.25
        pc2 = _PC(big)
        st_big = oracle_forward(dict(big, extra=feat), cams[0])
        rz._R_ESTIMATE[key0] = 1
        with torch.no_grad():                       # eval render: verified at once, redone with the exact size
            pkg = render(camg[0], pc2, _Pipe(), bg)
        np.testing.assert_array_equal(pkg["render"].cpu().numpy(), st_big["color"])
        assert rz._R_ESTIMATE[key0] == st_big["R"] and rz._CONFIG["async_binning"] is True
        rz._R_ESTIMATE[key0] = 1
        pc2._i["extra"].requires_grad_(True)        # training forward: the check fires before the backward kernels run
        pkg = render(camg[0], pc2, _Pipe(), bg)
        with pytest.raises(rz.BinningOverflow):
            pkg["seg_feature"].sum().backward()
        assert rz._R_ESTIMATE[key0] == st_big["R"] and rz._CONFIG["async_binning"] is True
        pkg = render(camg[0], pc2, _Pipe(), bg)     # re-run: sized by the corrected estimate, async again
        pkg["seg_feature"].sum().backward()
        np.testing.assert_array_equal(pkg["render"].detach().cpu().numpy(), st_big["color"])
    finally:
        rz.set_async_binning(False)
        rz._ASYNC_SLACK = 65536
        rz._R_ESTIMATE.clear(); rz._PENDING.clear(); rz._OVERFLOWED.clear()


def test_prefetched_chain_reads_its_count_back_behind_the_binning():
    """``prefetch_geometry`` on a side stream issues the asynchronous read-back of the view's instance count BEHIND the binning
    chain (not between the tile scan and the key scatter): an estimate that turns out too small is still caught before the
    backward kernels run - BinningOverflow, estimate corrected - and a chain that fits is consumed bit for bit."""
    import copy
    from helpers import oracle_forward
    from instascene_amd.render import prefetch
    sc, cams, inp = small_scene(P=900, F=8, W=64, H=48, seed=78)
    rz.set_mode("exact")
    rz.set_tracer(True)
    rz._R_ESTIMATE.clear(); rz._PENDING.clear(); rz._OVERFLOWED.clear()
    side = torch.cuda.Stream()
    try:
        rz.set_async_binning(True)
        cam = copy.deepcopy(cams[0]).to("cuda")
        bg = torch.zeros(3, device="cuda")
        feat = __import__("instascene_amd.contrastive", fromlist=["row_normalize"]).row_normalize(inp["extra"].cuda(), 1e-9).cpu()
        big = {k: (v.clone() if v is not None else None) for k, v in inp.items()}
        big["scales"] = big["scales"] * 30.0
        pc = _PC(big)
        st = oracle_forward(dict(big, extra=feat), cams[0])
        with torch.no_grad():
            render(cam, pc, _Pipe(), bg)                # first visit: sized exactly, the view's count verified
        key0 = next(iter(rz._R_ESTIMATE))
        assert rz._R_ESTIMATE[key0] == st["R"]
        pc._i["extra"].requires_grad_(True)
        # a chain that fits
        hits = rz.PREFETCH_HITS
        assert prefetch(cam, pc, _Pipe(), bg, stream=side)
        pkg = render(cam, pc, _Pipe(), bg)
        assert rz.PREFETCH_HITS == hits + 1
        pkg["seg_feature"].sum().backward()
        np.testing.assert_array_equal(pkg["render"].detach().cpu().numpy(), st["color"])
        # a chain sized from an estimate that is far too small
        slack, rz._ASYNC_SLACK = rz._ASYNC_SLACK, 0
        rz._R_ESTIMATE[key0] = 1
        assert prefetch(cam, pc, _Pipe(), bg, stream=side)
        pkg = render(cam, pc, _Pipe(), bg)
        with pytest.raises(rz.BinningOverflow):
            pkg["seg_feature"].sum().backward()
        assert rz._R_ESTIMATE[key0] == st["R"]
        rz._ASYNC_SLACK = slack
        pkg = render(cam, pc, _Pipe(), bg)
        pkg["seg_feature"].sum().backward()
        np.testing.assert_array_equal(pkg["render"].detach().cpu().numpy(), st["color"])
    finally:
        rz.set_async_binning(False)
        rz._ASYNC_SLACK = 65536
        rz._R_ESTIMATE.clear(); rz._PENDING.clear(); rz._OVERFLOWED.clear()
        torch.cuda.synchronize()


def _golden_cam(c, i):
    from instascene_amd import scenes
    W, H = (int(v) for v in c[f"wh{i}"])
    return scenes.Camera(W, H, float(c[f"fov{i}"][0]), float(c[f"fov{i}"][1]), torch.tensor(c[f"wvt{i}"]),
                         torch.tensor(c[f"proj{i}"]), torch.tensor(c[f"full{i}"]), torch.tensor(c[f"center{i}"]))


@pytest.mark.parametrize("i", range(3))
@pytest.mark.parametrize("ratio", [0, 1])
def test_fused_render_post_matches_reference_golden(golden_dir, i, ratio):
    """iso_render_post_forward against the reference's own render() post-processing outputs (tests/golden)."""
    import copy
    from instascene_amd.render import post_process
    z = np.load(os.path.join(golden_dir, "render_post.npz"))
    c = np.load(os.path.join(golden_dir, "cameras.npz"))
    cam = copy.deepcopy(_golden_cam(c, i)).to("cuda")
    out = post_process(cam, torch.tensor(z[f"c{i}_r{ratio}_allmap"]).cuda(), float(ratio))
    for k, v in out.items():
        assert_close(v.cpu().numpy(), z[f"c{i}_r{ratio}_{k}"], 2e-5, k)


@pytest.mark.parametrize("ratio", [0.0, 1.0, 0.3])
def test_fused_render_post_backward_matches_torch_autograd(golden_dir, ratio):
    """iso_render_post_backward against autograd of the torch restatement (float64), each output's gradient alone
    and all together; zero-alpha pixels (non-finite quotient) give 0 where torch gives NaN."""
    import copy
    from instascene_amd.render import post_process
    z = np.load(os.path.join(golden_dir, "render_post.npz"))
    c = np.load(os.path.join(golden_dir, "cameras.npz"))
    cam_cpu = _golden_cam(c, 1)
    cam_gpu = copy.deepcopy(_golden_cam(c, 1)).to("cuda")
    am = torch.tensor(z["c1_r0_allmap"]).clone()
    am[1].clamp_(min=1e-3)                       # keep the quotient finite for the comparison
    g = torch.Generator().manual_seed(3)
    keys = ["rend_alpha", "rend_normal", "rend_dist", "surf_depth", "surf_normal", "rend_depth", "rend_median_depth"]
    a = am.double().requires_grad_(True)
    from oracle import torch_ops
    ref = torch_ops.render_post(a, cam_cpu.world_view_transform.double(), cam_cpu.full_proj_transform.double(),
                                cam_cpu.image_width, cam_cpu.image_height, ratio)
    b = am.cuda().requires_grad_(True)
    got = post_process(cam_gpu, b, ratio)
    ups = {k: torch.randn(ref[k].shape, generator=g) for k in keys}
    for sel in [[k] for k in keys] + [keys]:
        a.grad = None; b.grad = None
        sum((ref[k] * ups[k].double()).sum() for k in sel).backward(retain_graph=True)
        sum((got[k] * ups[k].cuda()).sum() for k in sel).backward(retain_graph=True)
        assert_close(b.grad.cpu().numpy(), a.grad.float().numpy(), 2e-4, "render_post bwd " + "+".join(sel))
    # zero alpha: finite gradients
    am0 = am.clone(); am0[1, 5:9, 5:9] = 0.0; am0[0, 5:9, 5:9] = 0.0
    b0 = am0.cuda().requires_grad_(True)
    o0 = post_process(cam_gpu, b0, ratio)
    (o0["rend_depth"].sum() + o0["surf_normal"].sum()).backward()
    assert torch.isfinite(b0.grad).all()


@pytest.mark.parametrize("C,H,W", [(3, 70, 101), (3, 64, 96), (1, 33, 17)])
def test_hip_ssim_matches_torch_restatement_and_golden(golden_dir, C, H, W):
    """iso_ssim_forward/backward against the stacked depthwise torch restatement (pinned by the reference's golden on
    the host) in float64, and against the golden itself."""
    from instascene_amd import losses
    g = torch.Generator().manual_seed(H * W)
    a = torch.rand(C, H, W, generator=g)
    b = (a + 0.2 * torch.randn(C, H, W, generator=g)).clamp(0, 1)
    ra = a.double().requires_grad_(True)
    want = losses.ssim(ra, b.double())
    (want * 0.7).backward()
    ga = a.cuda().requires_grad_(True)
    got = losses.ssim(ga, b.cuda())
    (got * 0.7).backward()
    assert abs(float(got.detach()) - float(want.detach())) < 2e-6
    assert_close(ga.grad.cpu().numpy(), ra.grad.float().numpy(), 1e-4, "ssim grad")
    again = losses.ssim(a.cuda(), b.cuda())
    assert float(again) == float(got.detach())              # deterministic, also without the derivative maps
    # the reference's own outputs (value and the gradient of 0.8*L1 + 0.2*(1 - SSIM), tests/golden/make_goldens.py)
    z = np.load(os.path.join(golden_dir, "losses.npz"))
    img = torch.tensor(z["img"]).cuda().requires_grad_(True)
    gt = torch.tensor(z["gt"]).cuda()
    l1, ss = losses.l1_loss(img, gt), losses.ssim(img, gt)
    assert abs(float(ss.detach()) - float(z["ssim"])) < 1e-5
    (0.8 * l1 + 0.2 * (1.0 - ss)).backward()
    assert_close(img.grad.cpu().numpy(), z["grad"], 1e-4, "loss grad vs reference")
    # the fused photometric term (L1 inside the SSIM kernels) against the same reference outputs
    img2 = torch.tensor(z["img"]).cuda().requires_grad_(True)
    loss = losses.photometric_loss(img2, gt, 0.2)
    want_loss = 0.8 * float(z["l1"]) + 0.2 * (1.0 - float(z["ssim"]))
    assert abs(float(loss.detach()) - want_loss) < 2e-6
    loss.backward()
    assert_close(img2.grad.cpu().numpy(), z["grad"], 1e-4, "fused photometric grad vs reference")


@pytest.mark.parametrize("H,W,ln,ldist", [(70, 101, 0.05, 100.0), (64, 96, 0.05, 0.0), (33, 17, 0.0, 10.0)])
def test_train_loss_matches_the_composed_form(golden_dir, H, W, ln, ldist):
    """iso_train_loss_forward/backward (train.py:89-103 in three + one launches) against the same expression composed from
    torch ops in float64: value and the gradients of image, rend_normal, surf_normal, rend_dist; and the photometric part
    against the reference's golden."""
    from instascene_amd import losses
    g = torch.Generator().manual_seed(7 * H + W)
    img = torch.rand(3, H, W, generator=g)
    gt = (img + 0.2 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    rn = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0) * torch.rand(1, H, W, generator=g)
    sn = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0)
    rd = torch.rand(1, H, W, generator=g) * 0.01

    def composed(i, a, b, d):
        loss = 0.8 * (i - gt.to(i)).abs().mean() + 0.2 * (1.0 - losses.ssim(i, gt.to(i)))
        loss = loss + ldist * d.mean()
        return loss + ln * (1 - (a * b).sum(dim=0))[None].mean()

    ref = [t.double().requires_grad_(True) for t in (img, rn, sn, rd)]
    want = composed(*ref)
    (want * 1.7).backward()
    dev = [t.cuda().requires_grad_(True) for t in (img, rn, sn, rd)]
    got = losses.train_loss(dev[0], gt.cuda(), 0.2, dev[1], dev[2], ln, dev[3], ldist)
    (got * 1.7).backward()
    assert abs(float(got.detach()) - float(want.detach())) < 3e-6 * max(1.0, abs(float(want.detach())))
    assert_close(dev[0].grad.cpu().numpy(), ref[0].grad.float().numpy(), 1e-4, "dL/dimage")
    if ln != 0.0:
        assert_close(dev[1].grad.cpu().numpy(), ref[1].grad.float().numpy(), 1e-5, "dL/drend_normal")
        assert_close(dev[2].grad.cpu().numpy(), ref[2].grad.float().numpy(), 1e-5, "dL/dsurf_normal")
    else:
        assert dev[1].grad is None and dev[2].grad is None
    if ldist != 0.0:
        assert_close(dev[3].grad.cpu().numpy(), ref[3].grad.float().numpy(), 1e-5, "dL/drend_dist")
    else:
        assert dev[3].grad is None
    again = losses.train_loss(img.cuda(), gt.cuda(), 0.2, rn.cuda(), sn.cuda(), ln, rd.cuda(), ldist)
    assert float(again) == float(got.detach())              # deterministic
    # the photometric part alone == the reference's golden (value and gradient of 0.8*L1 + 0.2*(1 - SSIM))
    z = np.load(os.path.join(golden_dir, "losses.npz"))
    gi = torch.tensor(z["img"]).cuda().requires_grad_(True)
    loss = losses.train_loss(gi, torch.tensor(z["gt"]).cuda(), 0.2)
    assert abs(float(loss.detach()) - (0.8 * float(z["l1"]) + 0.2 * (1.0 - float(z["ssim"])))) < 2e-6
    loss.backward()
    assert_close(gi.grad.cpu().numpy(), z["grad"], 1e-4, "train_loss photometric grad vs reference")


def test_feature_adam_matches_torch_adam_and_emits_the_normalisation_chain():
    """iso_adam_rownorm2: parameters / moments like torch.optim.Adam(lr .025, eps 1e-15) over several steps, and the
    emitted (y, z) bit-identical to row_normalize_chain of the updated parameter; the chain stays differentiable."""
    from instascene_amd.contrastive import FeatureAdam, row_normalize, row_normalize_chain
    g = torch.Generator().manual_seed(12)
    x0 = torch.randn(3001, 32, generator=g)
    a = torch.nn.Parameter(x0.clone().cuda())
    b = torch.nn.Parameter(x0.clone().cuda())
    ref = torch.optim.Adam([a], lr=0.025, eps=1e-15)
    opt = FeatureAdam(b, lr=0.025, eps=1e-15, norm_eps=(1e-6, 1e-9))
    for it in range(6):
        gr = (torch.randn(3001, 32, generator=g) * (10.0 ** (-it))).cuda()
        gr[7] = 0.0
        a.grad, b.grad = gr.clone(), gr.clone()
        ref.step(); opt.step()
        assert_close(b.detach().cpu().numpy(), a.detach().cpu().numpy(), 2e-6, "adam param step %d" % it)
        st = ref.state[a]
        assert_close(opt.exp_avg.cpu().numpy(), st["exp_avg"].cpu().numpy(), 2e-6, "exp_avg")
        assert_close(opt.exp_avg_sq.cpu().numpy(), st["exp_avg_sq"].cpu().numpy(), 2e-6, "exp_avg_sq")
        y = opt.normalized_chain()
        y2 = row_normalize_chain(b.detach(), 1e-6, 1e-9)
        assert torch.equal(y.detach(), y2) and torch.equal(row_normalize(y, 1e-9).detach(), row_normalize(y2, 1e-9))
    # gradient through the given chain == gradient through the computed chain
    w = torch.randn(3001, 32, generator=g).cuda()
    b.grad = None
    y = opt.normalized_chain()
    ((y * w).sum() + (row_normalize(y, 1e-9) * w.flip(0)).sum()).backward()
    g_given = b.grad.clone(); b.grad = None
    yc = row_normalize_chain(b, 1e-6, 1e-9)
    ((yc * w).sum() + (row_normalize(yc, 1e-9) * w.flip(0)).sum()).backward()
    assert torch.equal(g_given, b.grad)
    # an in-place change of the parameter invalidates the emitted chain
    with torch.no_grad():
        b.mul_(2.0)
    y3 = opt.normalized_chain()
    assert torch.equal(y3.detach(), row_normalize_chain(b.detach(), 1e-6, 1e-9))


def test_sample_step_draws_uniformly_and_gathers_labels():
    """iso_sample_step: one launch for a step's index sampling.  Draws lie in the pools, labels equal the gathers torch
    would do, (seed, step) reproduces, different steps differ, and the draws are uniform over the pool (chi-square)."""
    import ctypes
    from instascene_amd._lib import check, lib
    g = torch.Generator().manual_seed(0)
    N, P, B = 5000, 3000, 8192
    pool2d = torch.randperm(N, generator=g)[:1000].sort().values.cuda()
    seg_a = torch.randint(0, 9, (N,), generator=g).cuda()
    seg_b = torch.randint(0, 9, (N,), generator=g).cuda()
    pool3d = torch.randperm(P, generator=g)[:700].cuda()
    lab3 = torch.randint(0, 9, (P,), generator=g).cuda()
    p = lambda t: ctypes.c_void_p(t.data_ptr())

    def draw(seed, step, n3=pool3d.numel()):
        out = torch.full((6 * B,), -7, dtype=torch.int64, device="cuda")
        check(lib().iso_sample_step(seed, step, B, pool2d.numel(), p(pool2d), p(seg_a), p(seg_b), n3, p(pool3d), p(lab3),
                                    p(out[:2 * B]), p(out[2 * B:3 * B]), p(out[3 * B:4 * B]), p(out[4 * B:5 * B]), p(out[5 * B:]),
                                    None), "iso_sample_step")
        return out[:2 * B], out[2 * B:3 * B], out[3 * B:4 * B], out[4 * B:5 * B], out[5 * B:]

    pix, la, lb, pk, l3 = draw(42, 7)
    assert bool(torch.isin(pix, pool2d).all()) and bool(torch.isin(pk, pool3d).all())
    assert torch.equal(la, seg_a[pix[:B]]) and torch.equal(lb, seg_b[pix[B:]]) and torch.equal(l3, lab3[pk])
    again = draw(42, 7)
    assert all(torch.equal(a, b) for a, b in zip((pix, la, lb, pk, l3), again))
    other = draw(42, 8)
    assert not torch.equal(pix, other[0]) and not torch.equal(pk, other[3])
    assert not torch.equal(pix, draw(43, 7)[0])
    # uniformity: 16384 draws over 1000 pool entries, chi-square with 999 degrees of freedom (mean 999, sd ~45)
    counts = torch.bincount(torch.searchsorted(pool2d, pix), minlength=1000).double()
    chi2 = float(((counts - 2 * B / 1000) ** 2 / (2 * B / 1000)).sum())
    assert 800 < chi2 < 1200, chi2
    # an empty 3-D pool leaves its outputs alone
    assert bool((draw(42, 7, n3=0)[3] == -7).all())


@pytest.mark.parametrize("P,F", [(5000, 32), (777, 16), (300, 64), (129, 40)])
def test_gather_rownorm_gives_the_stored_rows(P, F):
    """iso_gather_rownorm: rows of normalize(x) gathered from x itself == the same rows of the stored table (iso_rownorm2's
    y), bit for bit; out-of-range indices give zero rows.  And the placeholder FeatureAdam hands out with store_y = False
    may only be read through gather_rows inside a DeferredFeatureRows block."""
    import ctypes
    from instascene_amd._lib import check, lib
    from instascene_amd.contrastive import FeatureAdam, _RowNorm2, gather_rows
    g = torch.Generator().manual_seed(P)
    x = (torch.randn(P, F, generator=g) * 2.0).cuda()
    x[3] = 0.0
    y, _ = _RowNorm2.apply(x, 1e-6, 1e-9)
    idx = torch.randint(0, P, (2000,), generator=g).cuda()
    idx[5], idx[6] = -1, P
    out = torch.full((2000, F), 7.0, device="cuda")
    pp = lambda t: ctypes.c_void_p(t.data_ptr())
    check(lib().iso_gather_rownorm(2000, F, P, 1e-6, pp(x), pp(idx), pp(out), None), "iso_gather_rownorm")
    ok = (idx >= 0) & (idx < P)
    assert torch.equal(out[ok], y[idx[ok]]) and float(out[~ok].abs().max()) == 0.0
    if F % 4 == 0 and F <= 256:
        p = torch.nn.Parameter(x.clone())
        opt = FeatureAdam(p, lr=0.025, eps=1e-15)
        opt.store_y = False
        opt.leaf_mode = True
        opt.normalized = (p._version, None, y.clone())          # as the one-pass tail leaves it
        table = opt.normalized_chain()
        rows = gather_rows(table, idx[ok])
        assert torch.equal(rows.detach(), y[idx[ok]])
        with pytest.raises(RuntimeError):
            rows.sum().backward()                                # no DeferredFeatureRows block: refused, not silently lost


@pytest.mark.parametrize("nb,consider_negative,min_pixnum", [(2, False, 0), (4, True, 0), (3, False, 40)])
def test_contrastive_batch_variants(nb, consider_negative, min_pixnum):
    """Batches of 2 and 4, labels that include 0 as a class (consider_negative), small clusters dropped (min_pixnum):
    the batch equals the separate calls bit for bit."""
    from instascene_amd.contrastive import contrastive_loss_batch
    g = torch.Generator().manual_seed(nb * 7 + min_pixnum)
    N, F, K = 3000, 32, 40
    feats = [torch.randn(N, F, generator=g).cuda() for _ in range(nb)]
    labels = [torch.randint(0, K, (N,), generator=g).cuda() for _ in range(nb)]
    labels[0][:25] = K - 1                                        # a cluster that min_pixnum = 40 may drop
    pre = torch.nn.functional.normalize(torch.randn(K, F, generator=g), dim=1).cuda()
    predefs = [None if b % 2 == 0 else pre for b in range(nb)]
    w = [0.5 + 0.25 * b for b in range(nb)]
    a = [f.clone().requires_grad_(True) for f in feats]
    want = None
    for f, l, u, wi in zip(a, labels, predefs, w):
        t = contrastive_loss(f, l, predef_u_list=u, num_labels=K, consider_negative=consider_negative, min_pixnum=min_pixnum) * wi
        want = t if want is None else want + t
    want.backward()
    b = [f.clone().requires_grad_(True) for f in feats]
    got, _ = contrastive_loss_batch(b, labels, predefs, w, num_labels=K, consider_negative=consider_negative,
                                    min_pixnum=min_pixnum)
    got.backward()
    assert float(got.detach()) == float(want.detach())
    for x, y in zip(a, b):
        assert torch.equal(x.grad, y.grad)


def test_dist2_3nn_lds_bucketed_query_matches_brute_force():
    """The opt-in LDS-bucketed query (ISO_KNN_LDS=1, read once per process: hence a child process) gives the same bits
    as the brute-force oracle on uniform, clustered (dense cells beyond the LDS budget -> global fallback), planar and
    duplicated clouds."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle
from instascene_amd.knn import distCUDA2
for P, kind in [(5000, "uniform"), (20000, "clustered"), (3000, "planar"), (4000, "dup"), (3, "uniform"), (70000, "uniform")]:
    g = np.random.RandomState(P)
    if kind == "uniform": pts = g.rand(P, 3).astype(np.float32) * 3 - 1.5
    elif kind == "clustered":
        c = g.randn(8, 3) * 2
        pts = (c[g.randint(0, 8, P)] + g.randn(P, 3) * 0.02).astype(np.float32); pts[:5] += 50.0
    elif kind == "planar": pts = np.concatenate([g.rand(P, 2), np.zeros((P, 1))], 1).astype(np.float32)
    else:
        pts = g.rand(P // 2, 3).astype(np.float32); pts = np.concatenate([pts, pts], 0)
    want = oracle.dist2_3nn(pts)
    got = distCUDA2(torch.tensor(pts).cuda()).cpu().numpy()
    assert np.array_equal(got, want), (P, kind, np.abs(got - want).max())
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, ISO_KNN_LDS="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def _big_case(z, tag):
    """Regenerate the inputs of tests/golden/contrastive_loss_big.npz from its seed (make_goldens.py G3b) and check them against the
    checksums the fixture holds (a torch whose CPU generator draws differently must fail here, not in the comparison)."""
    seed = int(z[f"{tag}_seed"])
    Nb, F, K, pool_n, predef = (int(v) for v in z[f"{tag}_dims"])
    gg = torch.Generator().manual_seed(seed)
    pool = torch.randn(pool_n, F, generator=gg)
    pool_labels = torch.randint(0, K + 1, (pool_n,), generator=gg)
    idx = torch.randint(0, pool_n, (Nb,), generator=gg)
    predef_u = torch.nn.functional.normalize(torch.randn(K + 1, F, generator=gg), dim=1) if predef else None
    chk = z[f"{tag}_check"]
    got = [float(pool.double().sum()), float(pool.double().abs().sum()), float(idx.sum()), float(pool_labels.sum())]
    assert np.allclose(got, chk, rtol=1e-12, atol=1e-9), "the fixture's inputs cannot be regenerated from its seed on this torch"
    return pool, pool_labels, idx, predef_u, K


@pytest.mark.parametrize("tag", ["computed", "predef"])
def test_contrastive_at_the_reference_default_batch_matches_reference_golden(golden_dir, tag):
    """The reference's DEFAULT loss shape - sample_batchsize = 32 768 rows of seg_feat_dim = 16 drawn with replacement, K = 64
    (arguments/__init__.py:65,103; train_semantic.py:183-190; utils/contrastive_utils.py:18-73) - against the reference's own value
    and gradients (fixture generated by importing it).  At this size the similarity kernel takes its separate-reduce form
    (ck_similarity_small over 2 048 workgroups + ck_loss_reduce) and the row gradients of repeated draws are merged by the chained
    iso_rows_compact path; both are compared: the gradient w.r.t. the drawn rows and, through torch's index backward AND through
    compact_row_grads, the gradient w.r.t. the pool."""
    from instascene_amd.contrastive import compact_row_grads, contrastive_loss_batch
    z = np.load(os.path.join(golden_dir, "contrastive_loss_big.npz"))
    pool, pool_labels, idx, predef_u, K = _big_case(z, tag)
    pick, pick_pool = torch.tensor(z[f"{tag}_pick"]), torch.tensor(z[f"{tag}_pick_pool"])
    want = float(z[f"{tag}_loss"])
    p_ = pool.cuda().requires_grad_(True)
    idx_c = idx.cuda()
    f = p_[idx_c]
    f.retain_grad()
    lab = pool_labels[idx].cuda()
    pre = None if predef_u is None else predef_u.cuda()
    loss = contrastive_loss(f, lab, predef_u_list=pre)
    loss.backward()
    assert abs(float(loss.detach()) - want) <= 1e-4 * abs(want)
    gf, gp = f.grad, p_.grad

    def digest(g, rows, name):
        mx = float(z[f"{tag}_grad_pool_max"])
        assert_close(g[rows.cuda()].cpu().numpy(), z[f"{tag}_grad_{name}_rows"], 1e-3, f"{tag} {name} rows")
        cs = g.double().sum(0).cpu().numpy()
        assert np.abs(cs - z[f"{tag}_grad_{name}_colsum"]).max() <= 1e-3 * max(mx, np.abs(z[f"{tag}_grad_{name}_colsum"]).max())
        l1 = float(g.double().abs().sum())
        assert abs(l1 - float(z[f"{tag}_grad_{name}_l1"])) <= 1e-4 * float(z[f"{tag}_grad_{name}_l1"])
    digest(gf, pick, "f")
    digest(gp, pick_pool, "pool")
    # the trainer's path: repeated draws merged by iso_rows_compact (chains of repeats), no dense index backward
    slot, merged = compact_row_grads(idx_c, gf.contiguous(), pool.shape[0])
    dense = torch.zeros_like(p_.detach())
    rows = (slot >= 0).nonzero().squeeze(1)
    dense[rows] = merged[slot[rows].long()]
    digest(dense, pick_pool, "pool")
    assert torch.equal(dense, gp) or (dense - gp).abs().max().item() <= 1e-6 * float(z[f"{tag}_grad_pool_max"])
    # the batched entry the trainers use (three problems of this shape in one sequence of launches: 3 x 2 048 workgroups)
    fb = p_.detach()[idx_c].clone().requires_grad_(True)
    total, parts = contrastive_loss_batch([fb, fb, fb], [lab, lab, lab], [pre, pre, pre], [1.0, 0.5, 0.25], K + 1)
    total.backward()
    assert abs(float(total.detach()) - 1.75 * want) <= 1e-4 * abs(1.75 * want)
    assert abs(float(parts[1]) - 0.5 * want) <= 1e-4 * abs(want)
    assert_close(fb.grad[pick.cuda()].cpu().numpy() / 1.75, z[f"{tag}_grad_f_rows"], 1e-3, f"{tag} batched rows")
    from instascene_amd import contrastive as _c
    for t in _c._SLOT_TABLES.values():          # the persistent slot table: hand it back clean
        if t.slot.shape[0] == pool.shape[0]:
            t.slot.fill_(-1); t.dirty = False
