"""GPU tests of the train-step harnesses and of the tracer consumer."""
import copy
import math

import numpy as np
import pytest
import torch

from helpers import small_scene, oracle_forward

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from instascene_amd import scenes, rasterizer as rz
    from instascene_amd.harness import SegTrainer, RgbTrainer, PipelineParams
    from instascene_amd.render import render
    from instascene_amd.tracker import segmap_gaussians


def _scene(P=4000, F=16, W=128, H=96, seed=5):
    sc = scenes.synthetic_scene(P, F, seed, math.log(0.04))
    cams = scenes.ring_cameras(6, W, H)
    return sc, cams


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_seg_trainer_steps_and_is_deterministic(mode):
    rz.set_mode(mode)
    rz.set_tracer(False)
    outs = []
    for rep in range(2):
        sc, cams = _scene()
        tr = SegTrainer(sc, cams, device="cuda", sample_batchsize=2048, n_labels=12, use_class_feat=True, multiview=True,
                        sample_mv_frames=2, seed=3)
        p0 = tr.model._seg_feature.detach().clone()
        losses = [float(tr.step(it)) for it in range(11)]          # includes the multi-view branch at it == 0 and 10
        assert all(np.isfinite(losses))
        assert not torch.equal(tr.model._seg_feature.detach(), p0)
        outs.append((losses, tr.model._seg_feature.detach().clone()))
    # no float atomics anywhere on the path: two runs agree bit for bit
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])
    rz.set_mode("exact")
    rz.set_tracer(True)


def test_prefetched_geometry_pass_changes_nothing():
    """The data-parallel trainer issues the next view's geometry pass before the optimiser step (while RCCL sums the
    gradient).  With one rank the same code path must reproduce the plain loop bit for bit, and every forward after
    the first must find its geometry pass already issued."""
    rz.set_mode("fast")
    rz.set_tracer(False)
    rz.set_async_binning(True)
    try:
        outs = []
        for pf in (False, True):
            sc, cams = _scene()
            tr = SegTrainer(sc, cams, device="cuda", sample_batchsize=2048, n_labels=12, use_class_feat=True, seed=3,
                            prefetch_geometry=pf)
            tr.warm_view_caches()                        # every view rendered once: its verified instance count (blocking)
            tr.step(0)
            hits0 = rz.PREFETCH_HITS
            losses = [float(tr.step(it)) for it in range(1, 8)]
            outs.append((losses, tr.model._seg_feature.detach().clone(), rz.PREFETCH_HITS - hits0))
        assert outs[0][0] == outs[1][0]
        assert torch.equal(outs[0][1], outs[1][1])
        assert outs[0][2] == 0 and outs[1][2] == 7
    finally:
        rz.set_async_binning(False)
        rz.set_mode("exact")
        rz.set_tracer(True)


def test_cross_view_leg_gradients_collected_in_one_tensor_change_nothing():
    """The five extra views of the cross-view leg add their sampled backwards into ONE [P,F] tensor, touching only the rows
    their samples reach (``DeferredFeatureRows(collect_dense=True)``), instead of a dense reduction per view summed by
    autograd: same losses and parameters bit for bit."""
    rz.set_mode("fast")
    rz.set_tracer(False)
    try:
        outs = []
        for collect in (False, True):
            sc, cams = _scene()
            tr = SegTrainer(sc, cams, device="cuda", sample_batchsize=2048, n_labels=12, use_class_feat=True, multiview=True,
                            sample_mv_frames=3, seed=3)
            tr.collect_dense = collect
            losses = [float(tr.step(it)) for it in range(12)]          # the leg runs at it == 0 and 10
            outs.append((losses, tr.model._seg_feature.detach().clone()))
        assert outs[0][0] == outs[1][0]
        assert torch.equal(outs[0][1], outs[1][1])
    finally:
        rz.set_mode("exact")
        rz.set_tracer(True)


def test_view_cache_reproduces_the_plain_loop():
    """Opt-in cache of the per-view geometry pass + binning (frozen geometry): same losses and parameters bit for bit,
    every revisit of a view is a hit, and a second forward of a view whose backward is still outstanding bypasses it."""
    rz.set_mode("fast")
    rz.set_tracer(False)
    try:
        outs = []
        for gb in (0.0, 1.0):
            rz.set_view_cache(gb)
            sc, cams = _scene()
            tr = SegTrainer(sc, cams, device="cuda", sample_batchsize=2048, n_labels=12, use_class_feat=True, seed=3)
            hits0 = rz.VIEW_CACHE_HITS
            losses = [float(tr.step(it)) for it in range(14)]          # 6 views: 2 full cycles + 2
            outs.append((losses, tr.model._seg_feature.detach().clone(), rz.VIEW_CACHE_HITS - hits0))
        assert outs[0][0] == outs[1][0]
        assert torch.equal(outs[0][1], outs[1][1])
        assert outs[0][2] == 0 and outs[1][2] == 8
        # two forwards of one view before either backward: the second must not reuse (and overwrite) the first one's state
        m, cam = tr.model, tr.cams[0]
        a = render(cam, m, tr.pipe, tr.bg)["seg_feature"]
        h = rz.VIEW_CACHE_HITS
        b = render(cam, m, tr.pipe, tr.bg)["seg_feature"]
        assert rz.VIEW_CACHE_HITS in (h, h + 1)
        (a.sum() + 2.0 * b.sum()).backward()
        g_two = m._seg_feature.grad.clone(); m._seg_feature.grad = None
        rz.set_view_cache(0.0)
        m._seg_cache = None
        a = render(cam, m, tr.pipe, tr.bg)["seg_feature"]
        b = render(cam, m, tr.pipe, tr.bg)["seg_feature"]
        (a.sum() + 2.0 * b.sum()).backward()
        assert torch.equal(g_two, m._seg_feature.grad)
    finally:
        rz.set_view_cache(0.0)
        rz.set_mode("exact")
        rz.set_tracer(True)


def test_rgb_trainer_reduces_the_loss():
    rz.set_mode("fast")
    rz.set_tracer(False)
    sc, cams = _scene(P=3000, F=0, W=96, H=64, seed=9)
    # targets: renders of a perturbed copy of the scene
    ref = copy.deepcopy(sc)
    tr0 = RgbTrainer(ref, cams, [torch.zeros(3, 64, 96)] * len(cams), device="cuda")
    with torch.no_grad():
        targets = [render(c, tr0.model, tr0.pipe, tr0.bg)["render"].clone() for c in tr0.cams]
    g = torch.Generator().manual_seed(1)
    sc.features_dc = sc.features_dc + 0.3 * torch.randn(sc.features_dc.shape, generator=g)
    sc.opacity_logit = sc.opacity_logit - 0.5
    tr = RgbTrainer(sc, cams, targets, device="cuda")
    first = [float(tr.step(it)[0]) for it in range(6)]
    for it in range(6, 60):
        tr.step(it)
    last = [float(tr.step(it)[0]) for it in range(60, 66)]
    assert np.isfinite(first + last).all()
    assert np.mean(last) < 0.8 * np.mean(first)
    for grp in tr.opt.param_groups:
        for p in grp["params"]:
            assert torch.isfinite(p).all()
    rz.set_mode("exact")
    rz.set_tracer(True)


def test_segmap_gaussians_from_rendered_tracer():
    class PC:
        pass
    sc, cams, inp = small_scene(P=1500, F=4, W=96, H=64, seed=33, mu_s=math.log(0.08))
    cam = cams[0]
    st = oracle_forward(inp, cam, tracer=True)
    seg = scenes.voronoi_labels(96, 64, 5, 7)
    # reference logic on the oracle's tracer list
    want = {}
    grp = torch.tensor(st["tracer"]).long()
    lab = seg.reshape(-1)[grp[:, 1]]
    for m in torch.unique(lab).tolist():
        if m == 0:
            continue
        s = set(grp[lab == m, 0].tolist())
        if len(s) >= 20:
            want[m] = s
    rz.set_mode("exact")
    rz.set_tracer(True)
    from tests_support import pc_from_inputs
    pc = pc_from_inputs(inp)
    out = render(copy.deepcopy(cam).to("cuda"), pc, PipelineParams(), torch.zeros(3, device="cuda"))
    got, frame = segmap_gaussians(out["gau_related_pixels"], seg.cuda(), min_gaussians=20)
    assert sorted(got) == sorted(want)
    for m in want:
        assert set(got[m].tolist()) == want[m]
    assert set(frame.tolist()) == set(grp[:, 0].tolist())


def test_rgb_trainer_with_density_control():
    """train.py's loop incl. densification on a short schedule: the Gaussian count changes, parameters / Adam state /
    statistics stay consistent, the loss keeps going down and the rasterizer follows the new P."""
    rz.set_mode("fast")
    rz.set_tracer(False)
    try:
        sc, cams = _scene(P=1500, F=0, W=96, H=64, seed=11)
        ref = copy.deepcopy(sc)
        tr0 = RgbTrainer(ref, cams, [torch.zeros(3, 64, 96)] * len(cams), device="cuda")
        with torch.no_grad():
            targets = [render(c, tr0.model, tr0.pipe, tr0.bg)["render"].clone() for c in tr0.cams]
        g = torch.Generator().manual_seed(2)
        sc.xyz = sc.xyz + 0.02 * torch.randn(sc.xyz.shape, generator=g)
        sc.features_dc = sc.features_dc + 0.3 * torch.randn(sc.features_dc.shape, generator=g)
        tr = RgbTrainer(sc, cams, targets, device="cuda",
                        densify=dict(from_iter=3, until_iter=40, interval=4, opacity_reset_interval=16, grad_threshold=1e-5))
        counts, losses = [], []
        for it in range(24):
            loss, _ = tr.step(it)
            losses.append(float(loss))
            counts.append(tr.model._xyz.shape[0])
        assert len(set(counts)) > 1, "densification never changed the number of Gaussians"
        P = tr.model._xyz.shape[0]
        for grp in tr.opt.param_groups:
            p = grp["params"][0]
            assert p.shape[0] == P and p.requires_grad
            st = tr.opt.state.get(p)
            assert st is None or st["exp_avg"].shape == p.shape
        assert tr.densifier.denom.shape[0] == P and tr.densifier.max_radii2D.shape[0] == P
        assert all(np.isfinite(losses))
    finally:
        rz.set_mode("exact")
        rz.set_tracer(True)


def test_fused_feature_update_tracks_torch_adam():
    """SegTrainer with the fused Adam + normalisation pass vs torch.optim.Adam followed by the separate normalisation."""
    rz.set_mode("fast")
    rz.set_tracer(False)
    try:
        res = []
        for fused in (False, True):
            sc, cams = _scene()
            tr = SegTrainer(sc, cams, device="cuda", sample_batchsize=2048, n_labels=12, use_class_feat=True, seed=3,
                            fused_update=fused)
            losses = [float(tr.step(it)) for it in range(8)]
            res.append((losses, tr.model._seg_feature.detach().clone()))
        np.testing.assert_allclose(res[0][0], res[1][0], rtol=2e-4)
        d = (res[0][1] - res[1][1]).abs().max().item()
        assert d <= 2e-4 * res[0][1].abs().max().item(), d
    finally:
        rz.set_mode("exact")
        rz.set_tracer(True)


def test_sampled_path_matches_dense_path():
    """SegTrainer reads the rendered feature only at its sampled pixels.  With ``sampled_path`` the rasterizer gathers
    them itself and takes the [n, F] gradient back; without it the reference's formulation (index the dense map, dense
    gradient map in the backward) runs.  Same samples, same losses, the same parameters to rounding of the summation order."""
    rz.set_mode("exact")
    rz.set_tracer(False)
    outs = []
    for sp in (False, True):
        sc, cams = _scene()
        tr = SegTrainer(sc, cams, device="cuda", sample_batchsize=2048, n_labels=12, use_class_feat=True, multiview=True,
                        sample_mv_frames=2, seed=3, sampled_path=sp)
        losses = [float(tr.step(it)) for it in range(11)]
        outs.append((losses, tr.model._seg_feature.detach().clone()))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=2e-5)
    d = (outs[0][1] - outs[1][1]).abs().max().item()
    assert d <= 2e-4 * outs[0][1].abs().max().item(), d
    rz.set_tracer(True)


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_fused_tail_reproduces_the_separate_kernels(mode):
    """SegTrainer(fused_tail=True): everything between the blend backward and the next forward is one pass over the [P,F]
    rows (isr_feature_rows_step).  Same arithmetic as reduction + rownorm2 backward + FeatureAdam: identical parameters,
    including the iterations whose multi-view branch sends a dense gradient to the same leaves."""
    rz.set_mode(mode)
    rz.set_tracer(False)
    outs = []
    for ft in (False, True):
        sc, cams = _scene()
        tr = SegTrainer(sc, cams, device="cuda", sample_batchsize=2048, n_labels=12, use_class_feat=True, multiview=True,
                        sample_mv_frames=2, seed=3, fused_tail=ft)
        assert tr.fused_tail == ft
        losses = [float(tr.step(it)) for it in range(12)]
        outs.append((losses, tr.model._seg_feature.detach().clone(), tr.opt.exp_avg.clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    rz.set_mode("exact")
    rz.set_tracer(True)


def test_seg_trainer_reference_defaults():
    """The reference's default training configuration (arguments/__init__.py:65,103-104; train_semantic.py:115-129,143-172):
    sample_batchsize = 32 768, seg_feat_dim = 16, multi-view leg on.  Round 4's fused tail raised at this batch size (the sparse
    row gradient of the 3-D loss was limited to 16 384 samples); it must step, and reproduce the separate kernels bit for bit."""
    rz.set_mode("fast")
    rz.set_tracer(False)
    outs = []
    for ft in (False, True):
        sc, cams = _scene(P=20000, F=16, W=512, H=384)     # (<= 512 samples per tile: the sampled backward's fixed-order regime)
        tr = SegTrainer(sc, cams, device="cuda", sample_batchsize=32 * 1024, n_labels=12, use_class_feat=True, multiview=True,
                        sample_mv_frames=2, seed=3, fused_tail=ft)
        assert tr.fused_tail == ft and tr.batch == 32768
        p0 = tr.model._seg_feature.detach().clone()
        losses = [float(tr.step(it)) for it in range(12)]          # the multi-view branch runs at it == 0 and 10
        assert all(np.isfinite(losses)) and not torch.equal(tr.model._seg_feature.detach(), p0)
        outs.append((losses, tr.model._seg_feature.detach().clone(), tr.opt.exp_avg.clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    rz.set_mode("exact")
    rz.set_tracer(True)


def test_batched_losses_change_nothing():
    """SegTrainer(batched_losses=True) evaluates the step's three contrastive losses with one sequence of launches:
    same losses, same parameters as one call per loss."""
    rz.set_mode("exact")
    rz.set_tracer(False)
    outs = []
    for bl in (False, True):
        sc, cams = _scene()
        tr = SegTrainer(sc, cams, device="cuda", sample_batchsize=2048, n_labels=12, use_class_feat=True, multiview=True,
                        sample_mv_frames=2, seed=3, batched_losses=bl)
        losses = [float(tr.step(it)) for it in range(11)]
        outs.append((losses, tr.model._seg_feature.detach().clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])
    rz.set_tracer(True)


def test_multi_rank_form_of_the_tail_with_one_rank():
    """With several ranks the per-Gaussian tail stops at dL/dx (isr_feature_rows_step, grad_out), the gradient is
    all-reduced, and iso_adam_rownorm2 finishes.  Forced with one rank (the collective is then the identity) it must give
    the parameters of the one-pass tail bit for bit."""
    rz.set_mode("fast")
    rz.set_tracer(False)
    rz.set_async_binning(True)
    try:
        outs = []
        for split in (False, True):
            sc, cams = _scene()
            tr = SegTrainer(sc, cams, device="cuda", sample_batchsize=2048, n_labels=12, use_class_feat=True, seed=3)
            tr.split_tail = split
            losses = [float(tr.step(it)) for it in range(9)]
            outs.append((losses, tr.model._seg_feature.detach().clone(), tr.opt.exp_avg_sq.clone(), tr.opt.step_count))
        assert outs[0][0] == outs[1][0] and outs[0][3] == outs[1][3] == 9
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    finally:
        rz.set_async_binning(False)
        rz.set_mode("exact")
        rz.set_tracer(True)


def test_prime_leaves_no_trace():
    """SegTrainer.prime() runs steps for their side effects on the runtime (code objects, allocator, streams) and undoes
    their effect on the training state: the steps that follow are the steps of an unprimed trainer, bit for bit."""
    rz.set_mode("fast")
    rz.set_tracer(False)
    rz.set_async_binning(True)
    try:
        outs = []
        for primed in (False, True):
            sc, cams = _scene()
            tr = SegTrainer(sc, cams, device="cuda", sample_batchsize=2048, n_labels=12, use_class_feat=True, seed=3)
            tr.warm_view_caches()
            if primed:
                tr.prime()
            losses = [float(tr.step(it)) for it in range(6)]
            outs.append((losses, tr.model._seg_feature.detach().clone(), tr.opt.step_count))
        assert outs[0][0] == outs[1][0] and outs[0][2] == outs[1][2] == 6
        assert torch.equal(outs[0][1], outs[1][1])
    finally:
        rz.set_async_binning(False)
        rz.set_mode("exact")
        rz.set_tracer(True)


def test_feature_only_forward_changes_no_parameter():
    """Opt-in pipe.feature_only_forward: the blend kernel skips colour, the auxiliary maps and the tracer.  The feature map
    and everything the step trains on are computed by the same instructions, so losses and parameters are bit-identical
    to the full forward's; the skipped dict entries are None."""
    from instascene_amd.render import render
    rz.set_mode("fast")
    rz.set_tracer(True)
    try:
        outs = []
        for fo in (False, True):
            sc, cams = _scene()
            tr = SegTrainer(sc, cams, device="cuda", sample_batchsize=2048, n_labels=12, use_class_feat=True, seed=3)
            tr.pipe.feature_only_forward = fo
            losses = [float(tr.step(it)) for it in range(6)]
            with torch.no_grad():
                pkg = render(tr.cams[1], tr.model, tr.pipe, tr.bg)
            outs.append((losses, tr.model._seg_feature.detach().clone(), pkg))
        assert outs[0][0] == outs[1][0]
        assert torch.equal(outs[0][1], outs[1][1])
        full, fo = outs[0][2], outs[1][2]
        assert torch.equal(full["seg_feature"], fo["seg_feature"]) and torch.equal(full["radii"], fo["radii"])
        assert fo["render"] is None and fo["rend_normal"] is None and fo["gau_related_pixels"] is None
        assert set(fo.keys()) >= set(full.keys())
    finally:
        rz.set_mode("exact")


def test_gaussian_adam_matches_torch_adam():
    """optim.GaussianAdam (chain rule of the getters + Adam on the six train.py groups + next activations, one kernel)
    against autograd through the torch getters + torch.optim.Adam with the reference's learning rates and eps = 1e-15
    (scene/gaussian_model.py:239-249): same losses, and parameters / moments equal to fp32 rounding after 12 steps."""
    from instascene_amd.harness import RgbTrainer
    rz.set_mode("exact")
    rz.set_tracer(False)
    res = []
    for fused in (False, True):
        sc = scenes.synthetic_scene(3000, 0, 7, math.log(0.05))
        sc.seg_feature = None
        cams = scenes.ring_cameras(4, 96, 64)
        g = torch.Generator().manual_seed(1)
        targets = [torch.rand(3, 64, 96, generator=g) for _ in cams]
        tr = RgbTrainer(sc, cams, targets, device="cuda", fused_update=fused)
        losses = [float(tr.step(it)[0]) for it in range(12)]
        m = tr.model
        state = {}
        if fused:
            for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"):
                state[k] = (tr.opt.exp_avg[k].clone(), tr.opt.exp_avg_sq[k].clone())
        else:
            for grp in tr.opt.param_groups:
                st = tr.opt.state[grp["params"][0]]
                state[grp["name"]] = (st["exp_avg"].clone(), st["exp_avg_sq"].clone())
        res.append((losses, {k: getattr(m, a).detach().clone() for k, a in
                             dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity",
                                  scaling="_scaling", rotation="_rotation").items()}, state))
    (la, pa, sa), (lb, pb, sb) = res
    for x, y in zip(la, lb):
        assert abs(x - y) <= 2e-5 * abs(x), (la, lb)
    for k in pa:
        # a step moves a parameter by ~lr whatever the gradient's size; agreement is judged against that scale
        lr = dict(xyz=0.00016, f_dc=0.0025, f_rest=0.0025 / 20, opacity=0.05, scaling=0.005, rotation=0.001)[k]
        err = (pa[k] - pb[k].reshape(pa[k].shape)).abs().max().item()
        assert err <= 0.02 * lr * 12, (k, err)
        bad = ((pa[k] - pb[k].reshape(pa[k].shape)).abs() > 1e-3 * lr).float().mean().item()
        assert bad < 0.01, (k, bad)
        for j in (0, 1):
            ref, got = sa[k][j], sb[k][j].reshape(sa[k][j].shape)
            assert (ref - got).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-20, (k, j)
    rz.set_tracer(True)


@pytest.mark.gpu
def test_rgb_trainer_fused_loss_equals_the_composed_loss():
    """RgbTrainer's one-op loss (losses.train_loss: photometric + distortion + normal consistency) against the same
    step with the loss composed from torch ops: same loss values and parameters to fp32 rounding after a few steps."""
    from instascene_amd.harness import RgbTrainer
    rz.set_mode("exact")
    rz.set_tracer(False)
    res = []
    for fused in (False, True):
        sc = scenes.synthetic_scene(3000, 0, 7, math.log(0.05))
        sc.seg_feature = None
        cams = scenes.ring_cameras(4, 96, 64)
        g = torch.Generator().manual_seed(1)
        targets = [torch.rand(3, 64, 96, generator=g) for _ in cams]
        tr = RgbTrainer(sc, cams, targets, device="cuda", lambda_dist=100.0)
        tr.fused_loss = fused
        losses = [float(tr.step(it)[0]) for it in range(6)]
        res.append((losses, tr.model._xyz.detach().clone(), tr.model._scaling.detach().clone()))
    (la, xa, sa), (lb, xb, sb) = res
    for x, y in zip(la, lb):
        assert abs(x - y) <= 2e-5 * abs(x), (la, lb)
    assert (xa - xb).abs().max().item() <= 0.02 * 0.00016 * 6
    assert (sa - sb).abs().max().item() <= 0.02 * 0.005 * 6
    rz.set_tracer(True)


def test_plain_loop_trainer_runs_the_reference_iteration():
    """harness.PlainSegTrainer (bench.py's `dropin_plain`): the reference's iteration on render() + contrastive_loss() alone -
    steps, trains, covers every view once per epoch in random order, multi-view leg at iteration 0 and 10."""
    from instascene_amd.harness import PlainSegTrainer
    sc, cams = _scene()
    tr = PlainSegTrainer(sc, cams, device="cuda", sample_batchsize=1024, n_labels=12, sample_mv_frames=2, seed=2)
    p0 = tr.model._seg_feature.detach().clone()
    seen, losses = [], []
    for it in range(12):
        losses.append(float(tr.step(it)))
        seen.append(tr.last_view)
    assert all(np.isfinite(losses)) and not torch.equal(tr.model._seg_feature.detach(), p0)
    assert sorted(seen[:6]) == list(range(6)) and sorted(seen[6:]) == list(range(6)) and seen[:6] != list(range(6))


@pytest.mark.parametrize("F", [16, 32, 64])
def test_scaled_rows_reproduce_the_stored_normalised_table(F, monkeypatch):
    """SegTrainer's tail does not write z = normalize(normalize(param)) [P,F] any more but its two factors per row [P,2]; the blend
    applies them to the RAW rows it stages (isr_forward_render_scaled, FeatureAdam.store_z = False).  The same two multiplies in
    the same order: losses, parameters and moments bit-identical to the default run that stores the table (opt-in: ISR_SCALED_ROWS=1) - for the
    narrow (F = 16: colour / normal on the MFMAs' spare rows), the full (32) and the wide (64: two chunks per pass) staging paths,
    including the iterations whose multi-view leg renders further views from the same placeholder."""
    rz.set_mode("fast_reflists")
    rz.set_tracer(False)
    outs = []
    try:
        for scaled in ("0", "1"):
            monkeypatch.setenv("ISR_SCALED_ROWS", scaled)
            sc, cams = _scene(P=6000, F=F, W=256, H=192)
            tr = SegTrainer(sc, cams, device="cuda", sample_batchsize=2048, n_labels=12, use_class_feat=True, multiview=True,
                            sample_mv_frames=2, seed=5)
            assert tr.fused_tail and tr.opt.store_z == (scaled == "0")
            losses = [float(tr.step(it)) for it in range(12)]
            from instascene_amd.contrastive import _ScaledRows
            assert isinstance(tr.opt.normalized[2], _ScaledRows) == (scaled == "1")
            with torch.no_grad():           # an evaluation render after training: differentiates / normalises the usual way
                img = render(tr.cams[1], tr.model, tr.pipe, tr.bg)["seg_feature"].clone()
            outs.append((losses, tr.model._seg_feature.detach().clone(), tr.opt.exp_avg.clone(), tr.opt.exp_avg_sq.clone(), img))
        assert outs[0][0] == outs[1][0]
        for a, b in zip(outs[0][1:], outs[1][1:]):
            assert torch.equal(a, b)
    finally:
        rz.set_mode("exact")
        rz.set_tracer(True)


def test_c_entry_tail_without_the_3d_loss(monkeypatch):
    """lambda_3d = 0: the C-entry tail runs two losses, no row gather, no merged row gradients (NULL slot table) - same bits as
    the autograd path."""
    rz.set_mode("fast_reflists")
    rz.set_tracer(False)
    outs = []
    try:
        for c_tail in ("0", "1"):
            monkeypatch.setenv("ISR_C_TAIL", c_tail)
            sc, cams = _scene(P=5000, F=16, W=192, H=128)
            tr = SegTrainer(sc, cams, device="cuda", sample_batchsize=1024, n_labels=12, use_class_feat=True, seed=7, lambda_3d=0.0)
            taken = [0]
            if tr.c_tail:
                orig = tr._c_tail
                def counted(*a, _o=orig, **k):
                    taken[0] += 1
                    return _o(*a, **k)
                tr._c_tail = counted
            losses = [float(tr.step(it)) for it in range(9)]
            if tr.c_tail:
                assert taken[0] == 9
            outs.append((losses, tr.model._seg_feature.detach().clone(), tr.opt.exp_avg_sq.clone()))
        assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    finally:
        rz.set_mode("exact")
        rz.set_tracer(True)


@pytest.mark.parametrize("mode,scaled", [("fast_reflists", "0"), ("fast_reflists", "1"), ("exact", "0")])
def test_c_entry_tail_reproduces_the_autograd_step(mode, scaled, monkeypatch):
    """SegTrainer runs everything behind the blend of a common iteration - the three losses, their backward, the merge of the 3-D
    loss' row gradients, the sampled backward through the blend, the per-Gaussian tail - through ONE C entry (isr_seg_step_tail,
    include/instascene_rasterizer.h) instead of four autograd Functions and a backward pass of the engine.  The same launches in
    the same order: losses, parameters and both Adam moments bit-identical to the autograd path (ISR_C_TAIL=0), also across the
    iterations that must fall back to it (the multi-view leg at it % 10 == 0; the first visit of a view, whose 3-D pool is unknown)."""
    rz.set_mode(mode)
    rz.set_tracer(False)
    monkeypatch.setenv("ISR_SCALED_ROWS", scaled)
    outs = []
    try:
        for c_tail in ("0", "1"):
            monkeypatch.setenv("ISR_C_TAIL", c_tail)
            sc, cams = _scene(P=6000, F=32, W=256, H=192)
            tr = SegTrainer(sc, cams, device="cuda", sample_batchsize=2048, n_labels=12, use_class_feat=True, multiview=True,
                            sample_mv_frames=2, seed=5)
            assert tr.fused_tail and tr.c_tail == (c_tail == "1")
            taken = [0]
            if tr.c_tail:
                orig = tr._c_tail
                def counted(*a, _o=orig, **k):
                    taken[0] += 1
                    return _o(*a, **k)
                tr._c_tail = counted
            losses = [float(tr.step(it)) for it in range(23)]
            if tr.c_tail:
                assert taken[0] >= 12, taken          # most iterations (not it = 0, 10, 20, not a view's first visit)
            outs.append((losses, tr.model._seg_feature.detach().clone(), tr.opt.exp_avg.clone(), tr.opt.exp_avg_sq.clone(), tr.opt.step_count))
        assert outs[0][0] == outs[1][0] and outs[0][4] == outs[1][4] == 23
        for a, b in zip(outs[0][1:4], outs[1][1:4]):
            assert torch.equal(a, b)
    finally:
        rz.set_mode("exact")
        rz.set_tracer(True)


@pytest.mark.parametrize("c_tail", ["0", "1"])
def test_tail_gated_behind_the_key_scatter_changes_nothing(c_tail, monkeypatch):
    """ISR_GATE_TAIL (auto: on for long tails, BASELINE config 5): the per-Gaussian tail waits - on the device, through an event the
    library records right behind k_scatter (isr_forward_bin_event) - for the key scatter of the chain this step issued on the side
    stream.  Scheduling only: the same losses and parameters, through the C-entry tail and through the autograd path."""
    rz.set_mode("fast_reflists")
    rz.set_tracer(False)
    rz.set_async_binning(True)
    monkeypatch.setenv("ISR_C_TAIL", c_tail)
    outs = []
    try:
        for gate in ("0", "1"):
            monkeypatch.setenv("ISR_GATE_TAIL", gate)
            sc, cams = _scene(P=6000, F=32, W=256, H=192)
            tr = SegTrainer(sc, cams, device="cuda", sample_batchsize=2048, n_labels=12, use_class_feat=True, seed=5, prefetch_geometry=True)
            assert tr.gate_tail == (gate == "1")
            tr.warm_view_caches()
            losses = [float(tr.step(it)) for it in range(14)]
            if gate == "1":
                assert rz.LAST_SCATTER_EVENT is not None          # chains were issued on the side stream with the event
            outs.append((losses, tr.model._seg_feature.detach().clone(), tr.opt.exp_avg.clone()))
        assert outs[0][0] == outs[1][0]
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    finally:
        rz.set_scatter_gate(False)
        rz.set_async_binning(False)
        rz.set_mode("exact")
        rz.set_tracer(True)


@pytest.mark.parametrize("mode,ldist", [("fast_reflists", 0.0), ("fast_reflists", 100.0), ("exact", 0.0)])
def test_rgb_c_entry_tail_reproduces_the_autograd_step(mode, ldist, monkeypatch):
    """RgbTrainer runs everything behind render() - the loss (L1 + SSIM, normal consistency, distortion), its backward through the
    derived maps, the rasterizer's backward, the six-group Adam step - through ONE C entry (isr_rgb_step_tail) instead of three
    autograd Functions and a backward pass of the engine.  The same launches in the same order: losses, all six parameter groups
    and their Adam moments bit-identical to the autograd path (ISR_C_TAIL=0), with and without the distortion term."""
    rz.set_mode(mode)
    rz.set_tracer(False)
    outs = []
    try:
        for c_tail in ("0", "1"):
            monkeypatch.setenv("ISR_C_TAIL", c_tail)
            sc, cams = _scene(P=3000, F=0, W=160, H=112, seed=9)
            g = torch.Generator().manual_seed(2)
            targets = [torch.rand(3, 112, 160, generator=g) for _ in cams]
            tr = RgbTrainer(sc, cams, targets, device="cuda", lambda_dist=ldist)
            assert tr.fused_update and tr.c_tail == (c_tail == "1")
            taken = [0]
            if tr.c_tail:
                orig = tr._c_tail
                def counted(*a, _o=orig, **k):
                    taken[0] += 1
                    return _o(*a, **k)
                tr._c_tail = counted
            losses = [float(tr.step(it)[0]) for it in range(12)]
            if tr.c_tail:
                assert taken[0] == 12
            groups = [p.detach().clone() for gr in tr.opt.param_groups for p in gr["params"]]
            moments = [tr.opt.exp_avg[k].clone() for k in tr.opt.exp_avg] + [tr.opt.exp_avg_sq[k].clone() for k in tr.opt.exp_avg_sq]
            outs.append((losses, groups, moments, tr.opt.step_count))
        assert outs[0][0] == outs[1][0] and outs[0][3] == outs[1][3] == 12
        for a, b in zip(outs[0][1] + outs[0][2], outs[1][1] + outs[1][2]):
            assert torch.equal(a, b)
        assert any(not torch.equal(a, torch.zeros_like(a)) for a in outs[1][2])
    finally:
        rz.set_mode("exact")
        rz.set_tracer(True)
