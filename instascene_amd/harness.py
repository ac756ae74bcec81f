"""Train-step harness: the build's own counterpart of the reference's feature-training loop
(train_semantic.py:95-208) on synthetic scenes, plus the data-parallel
variant the north-star adds (one view per rank, RCCL sum all-reduce of the parameter gradient).

Per iteration, like the reference:
  1. pick a view (a seeded random permutation of the views per epoch - the reference's random pop from a refilled stack with
     the draws made up front, ``dist_utils.view_order`` - of which rank r takes entry ``it * world + r``);
  2. ``render()`` — full forward (RGB + depth + normal + F-dim feature), as the reference does even
     though only the feature is trained (:102);
  3. for each label map (``segmap``, and ``sorted_segmap`` iff class prototypes exist, :110-141):
     sample ``sample_batchsize`` labelled pixels with replacement, ``contrastive_loss`` * lambda_sv * {0.5|1};
  4. optional multi-view loss every 10th iteration over 5 consecutive views (:143-172; ``_multiview_loss``: batch drawn first,
     every view returns only its samples);
  5. 3-D loss on visible Gaussians' features vs their 3-D labels (:174-197), lambda 2.5e-6;
  6. backward; all-reduce(sum) of ``_seg_feature.grad`` across ranks; Adam(lr .025, eps 1e-15) (:203-208).
Geometry parameters are frozen, exactly like ``GaussianModel.training_setup`` does for this stage
(scene/gaussian_model.py:217-232).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from . import scenes
from .contrastive import contrastive_loss, contrastive_loss_batch, gather_rows, row_normalize_chain
from .dist_utils import (allreduce_bucket, allreduce_grads, allreduce_grads_async, allreduce_rows_async, row_ranges, view_for,
                         wait_all)
from .rasterizer import DeferredFeatureRows
from .render import prefetch, render


from .streams import main_stream, side_stream  # noqa: E402,F401  (one side / one high-priority stream per device)


def _unit_grad(owner, loss):
    """dL/dL = 1 from a tensor cached on the trainer (autograd otherwise fills a fresh one every step: one more launch in the
    step's chain)."""
    one = getattr(owner, "_one", None)
    if one is None or one.shape != loss.shape or one.device != loss.device or one.dtype != loss.dtype:
        one = owner._one = torch.ones_like(loss)
    return one


class PipelineParams:
    compute_cov3D_python = False
    convert_SHs_python = False
    depth_ratio = 1.0
    debug = False


def splat_to_world(xyz, scaling, scaling_modifier, rotation):
    """``GaussianModel.get_covariance`` (scene/gaussian_model.py:35-42,137-138): the 4x4 splat->world matrix of every
    surfel in row-vector storage - rows = the two scaled tangent axes, the normal, the centre; the quaternion (w,x,y,z) is
    normalised here like ``build_rotation`` does.  Only ``pipe.compute_cov3D_python`` reads it (render._precomputed_transforms)."""
    from .densify import rotation_matrices
    R = rotation_matrices(rotation)                              # [P,3,3], columns = local axes
    s = scaling * scaling_modifier
    out = xyz.new_zeros((xyz.shape[0], 4, 4))
    out[:, 0, :3] = R[:, :, 0] * s[:, 0:1]
    out[:, 1, :3] = R[:, :, 1] * s[:, 1:2]
    out[:, 2, :3] = R[:, :, 2]
    out[:, 3, :3] = xyz
    out[:, 3, 3] = 1.0
    return out


class SegGaussianModel:
    """The subset of the reference ``GaussianModel`` that ``render()`` and the loop touch
    (getters scene/gaussian_model.py:109-138; ``_seg_feature`` is the only trainable tensor)."""

    def __init__(self, scene: scenes.Scene, device, class_feat: Optional[torch.Tensor] = None):
        s = scene.to(device)
        self._xyz = s.xyz
        self._scaling = s.log_scale
        self._rotation = s.rot
        self._opacity = s.opacity_logit
        self._features_dc = s.features_dc
        self._features_rest = s.features_rest
        self._seg_feature = nn.Parameter(s.seg_feature.clone().requires_grad_(True)) if s.seg_feature is not None else None
        self.active_sh_degree = 3
        self.max_sh_degree = 3
        self.class_feat = class_feat
        self._features_cat = None
        self._seg_cache = None
        self._act_cache = {}

    get_xyz = property(lambda s: s._xyz)

    def get_covariance(self, scaling_modifier=1):
        return splat_to_world(self.get_xyz, self.get_scaling, scaling_modifier, self._rotation)

    def _frozen(self, name, src, fn):
        """Activation of a frozen parameter: loop-invariant in this stage, evaluated once (the reference re-runs
        exp / sigmoid / normalize on all P Gaussians every iteration, scene/gaussian_model.py:109-138)."""
        if src.requires_grad:
            return fn(src)
        hit = self._act_cache.get(name)
        if hit is None or hit[0] is not src or hit[1] != src._version:
            hit = (src, src._version, fn(src))
            self._act_cache[name] = hit
        return hit[2]

    get_scaling = property(lambda s: s._frozen("scaling", s._scaling, torch.exp))
    get_rotation = property(lambda s: s._frozen("rotation", s._rotation, torch.nn.functional.normalize))
    get_opacity = property(lambda s: s._frozen("opacity", s._opacity, torch.sigmoid))

    @property
    def get_features(self):
        # reference: torch.cat((dc, rest), dim=1) on every call (scene/gaussian_model.py:128-131).  Both parts are
        # frozen in this stage, so the concatenation is loop-invariant and done once.
        if self._features_dc.requires_grad or self._features_rest.requires_grad:
            return torch.cat((self._features_dc, self._features_rest), dim=1)
        if self._features_cat is None:
            self._features_cat = torch.cat((self._features_dc, self._features_rest), dim=1)
        return self._features_cat

    @property
    def get_seg_feature(self):
        if self._seg_feature is None:
            return None
        # called twice per step (render() and the 3-D loss): reuse the node while the parameter is unchanged
        key = (self._seg_feature._version, torch.is_grad_enabled())
        if self._seg_cache is None or self._seg_cache[0] != key:
            # eps 1e-6 here (scene/gaussian_model.py:122-125); render() re-normalises with 1e-9 — both in one pass, which
            # the fused optimiser step has already made when the parameter has not been touched since
            opt = getattr(self, "feature_optimizer", None)
            chain = opt.normalized_chain() if opt is not None else row_normalize_chain(self._seg_feature, 1e-6, 1e-9)
            self._seg_cache = (key, chain)
        return self._seg_cache[1]


def gram_schmidt(vectors: torch.Tensor) -> torch.Tensor:
    """scene/gaussian_model.py:161-167 (--gram_feat_3d class prototypes)."""
    out: List[torch.Tensor] = []
    for v in vectors:
        for u in out:
            v = v - torch.dot(v, u) * u
        out.append(v / (torch.norm(v) + 1e-9))
    return torch.stack(out)


class SegTrainer:
    def __init__(self, scene: scenes.Scene, cameras: List[scenes.Camera], device="cuda", sample_batchsize=8192,
                 n_labels=64, lambda_sv=1e-6, lambda_mv=1e-6, lambda_3d=2.5e-6, sample_mv_frames=5, use_class_feat=False,
                 multiview=False, seed=0, rank=0, world=1, prefetch_geometry=None, fused_update=None, sampled_path=True,
                 fused_tail=None, batched_losses=None, spatial_sort=True, fused_sampling=None):
        self.device = torch.device(device)
        # Gaussians stored in Z-order of their centres (once, here): neighbours in memory are neighbours on screen, which
        # the binning kernels' workgroup-level counter merging and every per-Gaussian gather rely on.  A pure relabelling
        # of rows: `self.order[k]` is the caller's index of row k (see `features_in_input_order`).
        self.order = None
        if spatial_sort:
            self.order = scenes.morton_order(scene.xyz)
            scene = scenes.spatially_sorted(scene, self.order)
        self.rank, self.world = rank, world
        self.sampled_path = bool(sampled_path)      # render(sample_pixels=...) instead of indexing the feature map
        self.batched_losses = (self.device.type == "cuda") if batched_losses is None else bool(batched_losses)
        # the next view's geometry pass + binning run on a side stream next to the rest of this step (_prefetch_next)
        import os as _os
        # (ISR_PREFETCH=0, diagnostic: every kernel of the step on ONE stream, i.e. its time without the side chain beside it)
        self.prefetch = (_os.environ.get("ISR_PREFETCH", "1") == "1") if prefetch_geometry is None else bool(prefetch_geometry)
        self._side = None
        # True: issue the next view's geometry pass + binning BEFORE this step's forward (they then run beside the blend
        # kernel, which is issue-bound and leaves the memory system idle) instead of behind it (beside the loss kernels, the
        # backward and the bandwidth-bound tail)
        self.draw_ahead = _os.environ.get("ISR_DRAW_AHEAD", "1") == "1"
        self.collect_dense = _os.environ.get("ISR_COLLECT_DENSE", "1") == "1"
        self.mv_chain_streams = int(_os.environ.get("ISR_MV_CHAIN_STREAMS", "3"))
        # ISR_PREFETCH_EARLY=1 / 0 select "early" / "after" (the two orders of rounds 2-3)
        self.prefetch_distance = max(1, int(_os.environ.get("ISR_PREFETCH_DISTANCE", "2")))
        _pe = _os.environ.get("ISR_PREFETCH_EARLY")
        self.prefetch_mode = _os.environ.get("ISR_PREFETCH_MODE", "behind" if _pe is None else ("early" if _pe == "1" else "after"))
        self.high_priority_main = _os.environ.get("ISR_MAIN_PRIORITY", "1") == "1"     # measured: 2.03 -> 1.995 ms per C3 step
        self.sharded_tail = _os.environ.get("ISR_SHARDED_TAIL", "0") == "1"    # opt-in (unmeasured on hardware): _tail_sharded
        # how the multi-rank tail sums dL/dparam: "rccl" (default: torch.distributed all-reduce in row ranges), "peer" (direct
        # reduce-scatter / all-gather over peer-mapped buffers) or "peer_compact" (only the rows this step touched) -
        # peer_exchange.PeerExchange, device-side phase flags; opt-in (ISR_EXCHANGE): never timed on a multi-GPU node
        self.exchange = _os.environ.get("ISR_EXCHANGE", "rccl")
        self._peer = None
        self.peer_check_every = max(1, int(os.environ.get("ISR_PEER_CHECK_EVERY", "50")))
        self.last_exchange = None
        self.phase_timing = False    # multi-rank tail: record per-phase device times of each step into self.last_phases
        self.last_phases = None
        self.split_tail = False      # tests: take the multi-rank form of the tail (dL/dx, all-reduce, Adam) with one rank
        self.stacked_losses = _os.environ.get("ISR_STACKED_LOSSES", "1") == "1"   # the two single-view losses read ONE [2B,F] input
        self.tail_chunks = 4         # row ranges of that form (all-reduce of one overlaps the kernels of the others)
        F = scene.seg_feature.shape[1]
        class_feat = None
        if use_class_feat:
            g = torch.Generator().manual_seed(seed + 17)
            class_feat = gram_schmidt(torch.rand(n_labels + 1, F, generator=g)).to(self.device)
        self.model = SegGaussianModel(scene, self.device, class_feat)
        self.labels3d = scene.labels3d.to(self.device).to(torch.int64).contiguous()     # iso_sample_step reads int64
        self.cams = [c.to(self.device) for c in cameras]
        self.pipe = PipelineParams()
        self.bg = torch.zeros(3, dtype=torch.float32, device=self.device)
        self.batch = sample_batchsize
        self.lsv, self.lmv, self.l3d = lambda_sv, lambda_mv, lambda_3d
        self.mv_frames, self.multiview = sample_mv_frames, multiview
        F_ = self.model._seg_feature.shape[1]
        if fused_update is None:
            fused_update = self.device.type == "cuda" and F_ % 4 == 0 and F_ <= 256
        if fused_update:       # Adam + the next forward's normalisation chain in one pass over [P,F]
            from .contrastive import FeatureAdam
            self.opt = FeatureAdam(self.model._seg_feature, lr=0.025, eps=1e-15, norm_eps=(1e-6, 1e-9))
            self.model.feature_optimizer = self.opt
        else:
            self.opt = torch.optim.Adam([{"params": [self.model._seg_feature], "lr": 0.025, "name": "seg_feature"}], lr=0.0,
                                        eps=1e-15, fused=self.device.type == "cuda")
        # everything between the blend backward and the next forward in one pass over the [P,F] rows: row reduction,
        # chain rule through both normalisations, Adam, next normalisations (FeatureAdam.step_rows)
        self.fused_tail = bool(fused_update and self.sampled_path) if fused_tail is None else bool(fused_tail)
        if self.fused_tail and not (fused_update and self.sampled_path):
            raise ValueError("fused_tail needs fused_update and sampled_path")
        if self.fused_tail:
            self.opt.store_y = False         # the step reads normalize(param) only through gather_rows (3-D loss)
            # ... and normalize(normalize(param)) only through the rasterizer's forward, which takes the raw table and the two
            # factors per row instead (one [P,F] stream less in the tail: FeatureAdam.store_z) - opt-in, ISR_SCALED_ROWS=1: bit-identical, but
            # measured no faster (C3 -2 %, C5 +-0: what the tail saves the blend's staging pays)
            self.opt.store_z = os.environ.get("ISR_SCALED_ROWS", "0") != "1"
        # the iteration behind the blend through one C entry where its shape allows (_c_tail_ok); ISR_C_TAIL=0: always autograd
        self.c_tail = os.environ.get("ISR_C_TAIL", "1") != "0"
        # the per-Gaussian tail (HBM-saturating) waits for the key scatter of the chain this step issued on the side stream
        # (rasterizer.set_scatter_gate): ISR_GATE_TAIL=1 / 0
        # "auto" (default): when the tail is long - P F >= 1.6e8, i.e. >= ~0.8 ms of [P,F] streams: BASELINE config 5 (4.20 / 4.09 ->
        # 4.06 / 4.02 ms, A/B/A/B on one box), not config 3, where the scatter is over before the tail starts and the wait costs 2 %
        _gt = os.environ.get("ISR_GATE_TAIL", "auto")
        _pf = int(self.model._seg_feature.shape[0]) * int(self.model._seg_feature.shape[1])
        self.gate_tail = _gt == "1" or (_gt == "auto" and _pf >= 160_000_000)
        if self.device.type == "cuda":
            from . import rasterizer as _rzg
            _rzg.set_scatter_gate(self.gate_tail)      # (process-wide: the most recently built trainer decides)
        self.view_seed = seed
        self.gen = torch.Generator(device=self.device).manual_seed(1000 + seed * 131 + rank)
        self.sample_seed = 1000 + seed * 131 + rank
        # all of a step's index sampling in one kernel (iso_sample_step; its own counter-based generator, so the samples
        # differ from the torch.randint ones of fused_sampling=False - same distribution)
        self.fused_sampling = (self.device.type == "cuda") if fused_sampling is None else bool(fused_sampling)
        # label maps are static per view: index the labelled pixels once (the reference re-derives the
        # boolean mask every iteration, train_semantic.py:118-125)
        self.n_labels = n_labels
        self._mv_pools = {}
        self.vis_pool = {}        # view -> indices of visible, labelled Gaussians (geometry is frozen: static per view)
        self.valid_idx = {}
        for i, c in enumerate(self.cams):
            if c.segmap is None:
                c.segmap = scenes.voronoi_labels(c.image_width, c.image_height, n_labels, 5000 + i, device=self.device)
                c.sorted_segmap = c.segmap
            # caller-supplied label maps come from image files (uint8 / int32, possibly strided views): the sampling kernel
            # (iso_sample_step) and the loss read contiguous int64
            same = c.sorted_segmap is c.segmap
            c.segmap = c.segmap.to(self.device).to(torch.int64).contiguous()
            c.sorted_segmap = c.segmap if (same or c.sorted_segmap is None) else \
                c.sorted_segmap.to(self.device).to(torch.int64).contiguous()
            self.valid_idx[i] = torch.nonzero(c.segmap.reshape(-1) > 0).reshape(-1)

    def features_in_input_order(self) -> torch.Tensor:
        """The trained ``[P,F]`` feature with rows in the order of the scene passed to the constructor."""
        f = self.model._seg_feature.detach()
        if self.order is None:
            return f.clone()
        out = torch.empty_like(f)
        out[self.order.to(f.device)] = f
        return out

    def warm_view_caches(self):
        """Per-view constants that the loop otherwise builds on the first visit of a view — the camera's ray table used
        by ``depth_to_normal`` (the reference rebuilds it on every call, utils/point_utils.py:10-27), the pool of visible
        labelled Gaussians for the 3-D loss, the binning-size estimate — computed up front with one untrained render per
        view, so that step times do not depend on how many views have been seen."""
        with torch.no_grad():
            for vi, cam in enumerate(self.cams):
                pkg = render(cam, self.model, self.pipe, self.bg)
                _ = pkg["surf_normal"]
                if self.l3d > 0 and vi not in self.vis_pool:
                    self.vis_pool[vi] = torch.nonzero(pkg["visibility_filter"] & (self.labels3d > 0)).reshape(-1)
        self.model._seg_cache = None

    def prime(self, steps: int = 2, next_it: int = 0):
        """Setup, not training: run ``steps`` iterations so that every kernel's code object is loaded, the caching
        allocator's pools have their steady-state size and the side stream exists — then put parameters, optimiser state and
        the sampling RNG back exactly as they were (so a benchmark's first warm-up step, iteration ``next_it``, is an ordinary
        step)."""
        p = self.model._seg_feature
        saved_p = p.detach().clone()
        gen_state = self.gen.get_state()
        fused = hasattr(self.opt, "exp_avg")
        if fused:
            saved_opt = (self.opt.exp_avg.clone(), self.opt.exp_avg_sq.clone(), self.opt.step_count)
        else:
            import copy
            saved_opt = copy.deepcopy(self.opt.state_dict())
        for it in range(steps):
            self.step(it)
        with torch.no_grad():
            p.copy_(saved_p)
        if fused:
            self.opt.exp_avg.copy_(saved_opt[0])
            self.opt.exp_avg_sq.copy_(saved_opt[1])
            self.opt.step_count = saved_opt[2]
            self.opt.normalized = None
        else:
            self.opt.load_state_dict(saved_opt)
        self.opt.zero_grad(set_to_none=True)
        self.gen.set_state(gen_state)
        self.model._seg_cache = None
        self._prefetch_next(next_it - 1)      # like every step does for its successor
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def _draw_samples(self, it, vi):
        """``(pix[2B], labels_a[B], labels_b[B], pick3d[B] | None, labels3d[pick3d] | None)`` of iteration ``it`` from ONE
        kernel (``iso_sample_step``) instead of two ``randint`` and six gathers; the 3-D part needs the view's pool of
        visible labelled Gaussians (``warm_view_caches`` or a first visit), else it is drawn later the torch way."""
        import ctypes
        from . import _hot
        from ._lib import check, lib
        B, dev = self.batch, self.device
        cam = self.cams[vi]
        pool2d = self.valid_idx[vi]
        pool3d = self.vis_pool.get(vi) if self.l3d > 0 else None
        n3 = 0 if pool3d is None else int(pool3d.numel())
        out = torch.empty(6 * B, dtype=torch.int64, device=dev)
        pix, la, lb, pick3d, lab3d = out[:2 * B], out[2 * B:3 * B], out[3 * B:4 * B], out[4 * B:5 * B], out[5 * B:]
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        with _hot.on_device(dev):
            check(lib().iso_sample_step(self.sample_seed, int(it), B, int(pool2d.numel()), p(pool2d), p(cam.segmap.reshape(-1)),
                                        p(cam.sorted_segmap.reshape(-1)), n3, p(pool3d) if n3 else None,
                                        p(self.labels3d) if n3 else None, p(pix), p(la), p(lb), p(pick3d), p(lab3d),
                                        _hot.stream_ptr(dev)), "iso_sample_step")
        return (pix, la, lb, pick3d if n3 else None, lab3d if n3 else None)

    def _sample_view_loss(self, vi, seg_feature, segmap, predef, weight):
        idx_pool = self.valid_idx[vi]
        if idx_pool.numel() == 0:
            return 0.0
        pick = torch.randint(0, idx_pool.numel(), (self.batch,), device=self.device, generator=self.gen)
        pix = idx_pool[pick]
        feats = seg_feature.reshape(seg_feature.shape[0], -1)[:, pix].T
        labels = segmap.reshape(-1)[pix]
        return contrastive_loss(feats, labels, predef_u_list=predef, num_labels=self.n_labels + 1) * (self.lsv * weight)

    def view_index(self, it):
        # a random permutation of the views per epoch (the reference's random pop, train_semantic.py:96-100), known ahead
        return view_for(it, self.rank, self.world, len(self.cams), seed=self.view_seed)

    def step(self, it: int):
        if self.high_priority_main and self.device.type == "cuda":
            # the step's own chain on a high-priority stream: the hardware favours its workgroups over those of the side
            # stream (the next view's binning), whose kernels otherwise slow the small loss kernels 2-3x
            ms, cur = main_stream(self.device), torch.cuda.current_stream(self.device)
            if cur == ms:                      # the caller already runs on it (``with trainer.stream_scope():``)
                return self._step_guarded(it)
            ms.wait_stream(cur)
            with torch.cuda.stream(ms):
                out = self._step_guarded(it)
            cur.wait_stream(ms)
            return out
        return self._step_guarded(it)

    def stream_scope(self):
        """``with trainer.stream_scope(): for it in ...: trainer.step(it)`` - the loop on the trainer's own high-priority
        stream, so that a step does not hop from the caller's stream to it and back (two cross-queue waits per step, ~30 us of
        idle chip at every step boundary: 1.947 -> 1.917 ms per C3 step); on exit the caller's stream waits for it."""
        import contextlib
        if not (self.high_priority_main and self.device.type == "cuda"):
            return contextlib.nullcontext()
        trainer = self

        class _Scope:
            def __enter__(self):
                self.ms, self.cur = main_stream(trainer.device), torch.cuda.current_stream(trainer.device)
                self.ms.wait_stream(self.cur)
                self.ctx = torch.cuda.stream(self.ms)
                self.ctx.__enter__()
                return trainer

            def __exit__(self, *exc):
                self.ctx.__exit__(*exc)
                self.cur.wait_stream(self.ms)
                return False

        return _Scope()

    def _unit_grad(self, loss):
        return _unit_grad(self, loss)

    def _step_guarded(self, it: int):
        from .rasterizer import BinningOverflow
        try:
            return self._step_once(it)
        except BinningOverflow:
            # the view needed more tile instances than its estimate allowed (async binning): the library has corrected the
            # estimate; nothing of this iteration has reached the parameters yet (the check precedes the backward kernels)
            self.opt.zero_grad(set_to_none=True)
            self.model._seg_cache = None
            return self._step_once(it)

    def _step_once(self, it: int):
        if not self.fused_tail:
            return self._step(it)
        # inside the step the normalised feature is a pair of autograd LEAVES (FeatureAdam.leaf_mode); outside it the
        # model differentiates down to the parameter as usual
        self.model._seg_cache = None
        self.opt.leaf_mode = True
        try:
            return self._step(it)
        finally:
            self.opt.leaf_mode = False
            self.opt.leaves = None
            self.model._seg_cache = None

    def _step(self, it: int):
        m = self.model
        vi = self.view_index(it)
        cam = self.cams[vi]
        merged = m.class_feat is not None and self.valid_idx[vi].numel() > 0
        pix = None
        drawn = None
        if merged and self.fused_sampling:
            # every index this step needs, from one kernel (iso_sample_step): pixels + their labels, 3-D picks + theirs
            ahead = getattr(self, "_drawn_ahead", None)
            self._drawn_ahead = None
            # (drawn one step early, behind the previous forward: the draw is a function of (seed, it, view) alone, so it
            # need not sit between the previous step's tail and this forward)
            drawn = ahead[2] if (ahead is not None and ahead[0] == it and ahead[1] == vi) else self._draw_samples(it, vi)
            pix = drawn[0]
        elif merged:
            # the pixels do not depend on the render: choose them first and let the rasterizer hand back the features at
            # those pixels (both single-view sample sets in ONE list; no dense dL/dfeature map in the backward)
            pool = self.valid_idx[vi]
            pick = torch.randint(0, pool.numel(), (2 * self.batch,), device=self.device, generator=self.gen)
            pix = pool[pick]
        # When does the side stream's chain (the NEXT view's geometry pass + binning) become runnable?
        #   "early"  before this forward is enqueued: its first kernels take the chip's LDS ahead of the blend
        #   "behind" (default) as soon as the blend has been ENQUEUED - an event recorded before it, the side launches after
        #            it: the blend's workgroups are placed first and the chain fills in at its tail and under the loss
        #            kernels
        #   "after"  when the blend has completed
        mode = self.prefetch_mode
        if mode == "early":
            self._prefetch_next(it)
        gate = None
        if mode == "behind" and self.prefetch and self.device.type == "cuda":
            gate = torch.cuda.Event()
            gate.record()
        pkg = render(cam, m, self.pipe, self.bg, sample_pixels=pix if self.sampled_path else None)
        if mode != "early":
            self._prefetch_next(it, after=gate)
        if merged and self.fused_sampling and self.draw_ahead:
            vn = self.view_index(it + 1)
            if self.valid_idx[vn].numel() > 0 and (self.l3d <= 0 or self.vis_pool.get(vn) is not None):
                self._drawn_ahead = (it + 1, vn, self._draw_samples(it + 1, vn))
        gate_ev = None
        if self.gate_tail and self.prefetch and self.device.type == "cuda" and self.world == 1:
            from . import rasterizer as _rzg
            gate_ev = _rzg.LAST_SCATTER_EVENT
        if self._c_tail_ok(it, merged, drawn, pkg):
            return self._c_tail(pkg, drawn, gate_ev)
        seg_feature = pkg["seg_feature"]
        # the step's prototype-contrastive losses, as (features, labels, predefined prototypes, weight)
        problems = []
        if merged:
            feats = pkg["sampled_seg_feature"] if self.sampled_path else seg_feature.reshape(seg_feature.shape[0], -1)[:, pix].T
            fa, fb = feats.split(self.batch)          # one cat in the backward instead of two zero-fill + copy + add
            stacked_feats = feats if (self.sampled_path and feats.shape[0] == 2 * self.batch) else None
            if drawn is not None:
                la, lb = drawn[1], drawn[2]
            else:
                la = cam.segmap.reshape(-1)[pix[:self.batch]]
                lb = cam.sorted_segmap.reshape(-1)[pix[self.batch:]]
            problems.append((fa, la, None, self.lsv * 0.5))
            problems.append((fb, lb, m.class_feat, self.lsv * 1.0))
            loss = None
        else:
            loss = self._sample_view_loss(vi, seg_feature, cam.segmap, None, 0.5)
        if self.l3d > 0:
            # reference :175-190 materialises feature[visibility_filter] ([V,F]) and then samples; sampling the
            # visible & labelled Gaussians first and gathering only the batch rows draws from the same distribution
            pool = self.vis_pool.get(vi)
            if pool is None:
                pool = torch.nonzero(pkg["visibility_filter"] & (self.labels3d > 0)).reshape(-1)
                self.vis_pool[vi] = pool
            if pool.numel() > 0:
                if drawn is not None and drawn[3] is not None:
                    pick, lab3d = drawn[3], drawn[4]
                else:
                    pick = pool[torch.randint(0, pool.numel(), (self.batch,), device=self.device, generator=self.gen)]
                    lab3d = self.labels3d[pick]
                rows3d = gather_rows(m.get_seg_feature, pick) if self.fused_tail else m.get_seg_feature[pick]
                problems.append((rows3d, lab3d, m.class_feat, self.l3d))
        if problems:
            K = self.n_labels + 1
            same = all(f.shape == problems[0][0].shape and (u is None or u.shape[0] == K) for f, _, u, _ in problems)
            if self.batched_losses and len(problems) > 1 and same:
                # one sequence of launches for all of them (iso_contrastive_forward_batch): each loss is ~8 kernels of a
                # few microseconds, i.e. launch-bound
                fl = [q[0] for q in problems]
                stk = 0
                if merged and self.stacked_losses and stacked_feats is not None and fl[0] is fa and fl[1] is fb:
                    fl, stk = [stacked_feats] + fl[2:], 2      # the render's sampled rows as ONE input: no cat in the backward
                part = contrastive_loss_batch(fl, [q[1] for q in problems], [q[2] for q in problems],
                                              [q[3] for q in problems], num_labels=K, stacked=stk)[0]
                loss = part if loss is None else loss + part
            else:
                for f, l, u, w in problems:
                    term = contrastive_loss(f, l, predef_u_list=u, num_labels=K) * w
                    loss = term if loss is None else loss + term
        if self.multiview and self.lmv > 0 and it % 10 == 0:
            loss = loss + self._multiview_loss(it, vi)
        if self.fused_tail:
            # (every render of the step differentiates the same z leaf: the cross-view leg's sampled backwards add into one
            # [P,F] tensor, sink.dense, which the tail takes as the dense part of dL/dz)
            with DeferredFeatureRows(collect_dense=self.collect_dense) as sink:
                loss.backward(self._unit_grad(loss))
            if self.world == 1 and not self.split_tail:
                if gate_ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(gate_ev)
                self.opt.step_rows(sink.rows, row_grads=sink.row_grads, dense=sink.dense)
                m._seg_cache = None
                return loss.detach()
            if self.sharded_tail and m._seg_feature.shape[0] % self.world == 0:
                self._tail_sharded(sink)
            elif self.exchange in ("peer", "peer_compact") and self.world > 1 and self.device.type == "cuda":
                self._tail_with_peer_exchange(sink)
            else:
                self._tail_with_allreduce(sink)
            m._seg_cache = None
            return loss.detach()
        loss.backward()
        if self.prefetch:
            wait_all(allreduce_grads_async([m._seg_feature], self.world))
        else:
            allreduce_grads([m._seg_feature], self.world)
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        m._seg_cache = None          # the graph of this step is gone
        return loss.detach()

    # -- everything behind the blend through ONE C entry (isr_seg_step_tail) instead of the autograd graph -----------------
    def _c_tail_ok(self, it, merged, drawn, pkg) -> bool:
        """The common iteration - one view, both single-view losses on the render's sampled pixels, the 3-D loss on drawn rows,
        one rank - has a fixed shape: its losses, their backward, the sampled backward through the blend and the per-Gaussian
        tail are then ONE host call (isr_seg_step_tail: the same launches in the same order, bit-identical parameters) instead of
        four autograd Functions, a graph and a backward pass of the autograd engine (~0.8 ms of host work per step).  Every other
        iteration (the multi-view leg, a view whose 3-D pool is not known yet, several ranks) takes the autograd path below."""
        if not (self.c_tail and self.fused_tail and self.world == 1 and not self.split_tail and merged and drawn is not None
                and self.sampled_path and self.batched_losses and self.stacked_losses and self.device.type == "cuda"):
            return False
        if self.multiview and self.lmv > 0 and it % 10 == 0:
            return False
        if self.l3d > 0 and drawn[3] is None:
            return False
        m = self.model
        if m.class_feat is None or m.class_feat.shape[0] != self.n_labels + 1 or not m.class_feat.is_cuda:
            return False
        sampled = pkg.get("sampled_seg_feature") if hasattr(pkg, "get") else None
        node = getattr(sampled, "grad_fn", None)
        return (sampled is not None and node is not None and hasattr(node, "num_rendered") and sampled.shape[0] == 2 * self.batch
                and self.opt.leaves is not None)

    def _c_tail(self, pkg, drawn, gate_ev=None):
        import ctypes
        from . import _hot, rasterizer as _rz
        from ._lib import check, lib
        from .contrastive import _ScaledRows, _slot_consumed, _slot_table
        L = lib()
        m, opt, dev, B = self.model, self.opt, self.device, self.batch
        p = opt.param
        P, F = p.shape
        K = self.n_labels + 1
        sampled = pkg["sampled_seg_feature"]
        node = sampled.grad_fn                      # the rasterizer's forward context: its state buffers, its instance count
        saved = node.saved_tensors
        geom, binning, img = saved[8], saved[9], saved[10]
        rs, R, mode = node.raster_settings, int(node.num_rendered), int(node.mode)
        W, H = int(rs.image_width), int(rs.image_height)
        _rz._verify_pending(geom.data_ptr())        # (async binning: BinningOverflow before anything reaches the parameters)
        pix, la, lb, pick3d, lab3d = drawn
        has3d = self.l3d > 0 and pick3d is not None
        nb = 3 if has3d else 2
        ws = getattr(self, "_tail_ws", None)
        if ws is None or ws["key"] != (P, F, K, B, dev):
            one = L.iso_contrastive_scratch_bytes(B, F, K)
            ws = self._tail_ws = {
                "key": (P, F, K, B, dev), "state_bytes": 3 * one,
                "state": torch.empty(3 * one, dtype=torch.uint8, device=dev),
                "rows3d": torch.empty((B, F), dtype=torch.float32, device=dev),
                "grads": torch.empty((3 * B, F), dtype=torch.float32, device=dev),
                "merged": torch.empty((B, F), dtype=torch.float32, device=dev),
                "chain": torch.empty(B, dtype=torch.int32, device=dev),
                "one": torch.ones(1, dtype=torch.float32, device=dev)}
        sampled_c = sampled.detach()
        if not sampled_c.is_contiguous():
            sampled_c = sampled_c.contiguous()
        lossbuf = torch.empty(4, dtype=torch.float32, device=dev)
        scratch = _rz._workspace(lambda c: L.isr_backward_sampled_scratch_bytes(c, F, 2 * B, W, H), R, dev)
        scaled = opt._scaled_rows_wanted()
        z = None if scaled else torch.empty_like(p.data)
        zs = opt._scale_buffer() if scaled else None
        table = _slot_table(P, dev) if has3d else None
        opt.step_count += 1
        ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        with _hot.on_device(dev):
            check(L.isr_seg_step_tail(
                P, F, K, B, W, H, mode, R, ptr(geom), ptr(binning), ptr(img), ptr(pix), ptr(sampled_c), ptr(la), ptr(lb),
                ptr(pick3d) if has3d else None, ptr(lab3d) if has3d else None, ptr(m.class_feat),
                float(self.lsv * 0.5), float(self.lsv * 1.0), float(self.l3d) if has3d else 0.0, 1000.0,
                ptr(p.data), ptr(opt.exp_avg), ptr(opt.exp_avg_sq), ptr(z), ptr(zs), opt.lr, float(opt.betas[0]), float(opt.betas[1]),
                opt.eps, max(1, opt.step_count), float(opt.norm_eps[0]), float(opt.norm_eps[1]),
                ptr(table.slot) if has3d else None, (0 if table.dirty else 1) if has3d else 1,
                ptr(ws["state"]), ws["state_bytes"], ptr(ws["rows3d"]), ptr(ws["grads"]), ptr(ws["merged"]), ptr(ws["chain"]),
                ptr(scratch), scratch.numel(), ptr(ws["one"]), ptr(lossbuf), ctypes.c_void_p(lossbuf.data_ptr() + 4 * nb),
                ctypes.c_void_p(gate_ev.cuda_event) if gate_ev is not None else None, _hot.stream_ptr(dev)), "isr_seg_step_tail")
        if has3d:
            table.dirty, table.covered = True, 0
            _slot_consumed(table.slot, P, P)
        torch.autograd.graph.increment_version(p)
        opt.normalized = (p._version, None, _ScaledRows(zs) if scaled else z)
        opt.leaves = None
        m._seg_cache = None
        return lossbuf[nb]

    def _multiview_loss(self, it, vi):
        """The cross-view leg (train_semantic.py:143-172): ``mv_frames`` consecutive views rendered with gradients, one
        batch sampled uniformly from their labelled pixels, one loss against the class prototypes.  The reference stacks the
        five dense ``[F,H,W]`` maps and mask-gathers them (1.3 GB copied, 1.2 GB gathered, the same again zero-filled and
        scattered in the backward); here the batch is drawn FIRST - how many of the B samples fall into each view is a host-
        side multinomial draw with probabilities proportional to the views' labelled-pixel counts, the pixels themselves are
        drawn on the device: the same distribution as uniform over the union - and every view hands back only its samples
        (``render(sample_pixels=)``).  The next view's geometry pass is issued on the side stream while one renders."""
        import numpy as np
        m = self.model
        n = len(self.cams)
        views = self._multiview_views(vi)
        pools = [self._sorted_pool(k) for k in views]
        sizes = np.array([p.numel() for p in pools], dtype=np.float64)
        if sizes.sum() == 0:
            return 0.0
        counts = np.random.RandomState((self.sample_seed * 7919 + it) & 0x7FFFFFFF).multinomial(self.batch, sizes / sizes.sum())
        feats, labs = [], []
        lanes = self.mv_chain_streams
        if lanes > 0:
            self._issue_leg_chains(views, counts)
        for j, (k, pool, nk) in enumerate(zip(views, pools, counts)):
            if lanes <= 0 and self.prefetch and self.device.type == "cuda" and j + 1 < len(views) and counts[j + 1] > 0:
                if self._side is None:
                    self._side = side_stream(self.device)
                prefetch(self.cams[views[j + 1]], m, self.pipe, self.bg, stream=self._side)
            if nk == 0:
                continue
            pix = pool[torch.randint(0, pool.numel(), (int(nk),), device=self.device, generator=self.gen)]
            if self.sampled_path:
                p2 = render(self.cams[k], m, self.pipe, self.bg, sample_pixels=pix, defer_rows=False)
                feats.append(p2["sampled_seg_feature"])
            else:
                sf = render(self.cams[k], m, self.pipe, self.bg)["seg_feature"]
                feats.append(sf.reshape(sf.shape[0], -1)[:, pix].T)
            labs.append(self.cams[k].sorted_segmap.reshape(-1)[pix])
        return contrastive_loss(torch.cat(feats, dim=0), torch.cat(labs), predef_u_list=m.class_feat,
                                num_labels=self.n_labels + 1) * self.lmv

    def _multiview_views(self, vi):
        """The window of ``sample_mv_frames`` consecutive views of the cross-view leg (train_semantic.py:146 draws its start at
        random; here a function of the step's view, so that the leg's chains can be issued ahead).  With no more views than the
        window the reference's draw has no valid start either: the window is clamped to the views there are."""
        n = len(self.cams)
        k = min(self.mv_frames, n)
        first = (vi + 1) % max(1, n - k + 1)
        return list(range(first, first + k))

    def _issue_leg_chains(self, views, counts=None):
        """All of the cross-view leg's binning chains at once, spread over ``mv_chain_streams`` side streams: a chain cannot
        run under a blend (the blend's waves hold every register), so one issued per view, inside the leg, is exposed in
        full; two or three of them next to each other take hardly longer than one (latency- and atomic-bound kernels)."""
        lanes = self.mv_chain_streams
        if lanes <= 0 or not self.prefetch or self.device.type != "cuda":
            return
        from .streams import extra_side_stream
        if self._side is None:
            self._side = side_stream(self.device)
        for j, k in enumerate(views):
            if counts is None or counts[j] > 0:
                st = self._side if j % lanes == 0 else extra_side_stream(self.device, j % lanes)
                prefetch(self.cams[k], self.model, self.pipe, self.bg, stream=st)

    def _sorted_pool(self, k):
        """Flat indices of the pixels of view ``k`` with a positive ``sorted_segmap`` label (static per view: cached)."""
        cam = self.cams[k]
        if cam.sorted_segmap is cam.segmap:
            return self.valid_idx[k]
        pool = self._mv_pools.get(k)
        if pool is None:
            pool = self._mv_pools[k] = torch.nonzero(cam.sorted_segmap.reshape(-1) > 0).reshape(-1)
        return pool

    def _tail_sharded(self, sink):
        """Several ranks, opt-in (``sharded_tail``; P divisible by the world size): reduce-scatter of dL/dparam, Adam on THIS
        rank's contiguous shard of the rows only, all-gather of the updated parameter rows, local re-normalisation of the rows
        that arrived.  The same bytes on the links as the all-reduce (its two halves), but each rank runs the optimiser pass on
        1/W of the table (and needs the Adam moments of its shard only): per rank ~5 of the 8 [P,F] streams of the replicated
        tail disappear.  Replicas stay bit-identical (every row is computed once, by its owner)."""
        from .dist_utils import all_gather_rows, reduce_scatter_rows, shard_rows
        opt, p = self.opt, self.model._seg_feature
        P = p.shape[0]
        tail = opt.begin_tail(sink.rows, sink.row_grads, sink.dense)
        if tail is None:
            p.grad = torch.zeros_like(p.data)
        else:
            opt.tail_gradient(tail, 0, P)
        r0, r1 = shard_rows(P, self.rank, self.world)
        mine = reduce_scatter_rows(p.grad, self.rank, self.world)
        opt.begin_step()
        opt.step_range(r0, r1, grad_rows=mine)
        all_gather_rows(p.data, self.rank, self.world)
        opt.renormalize_range(0, r0)
        opt.renormalize_range(r1, P)
        opt.end_step()
        opt.zero_grad(set_to_none=True)

    def _tail_with_peer_exchange(self, sink):
        """Several ranks, opt-in (``exchange`` = "peer" / "peer_compact"): dL/dparam is reduced into a persistent ``[P,F]`` buffer
        that all ranks have mapped, summed by the direct exchange (peer_exchange.PeerExchange: generation counters in device
        memory, no host round trip; every row summed in rank order, so the replicas stay bit-identical), then Adam on every rank.
        "peer_compact" exchanges only the rows this rank's step touched (a row is touched when its gradient is non-zero)."""
        from .peer_exchange import PeerExchange
        opt, p = self.opt, self.model._seg_feature
        P, F = p.shape
        if self._peer is None:
            self._xgrad = torch.zeros(P, F, dtype=torch.float32, device=self.device)
            self._peer = PeerExchange(self._xgrad, rows=(P, F) if self.exchange == "peer_compact" else None)
        tail = opt.begin_tail(sink.rows, sink.row_grads, sink.dense)
        p.grad = self._xgrad
        if tail is None:
            self._xgrad.zero_()
        else:
            opt.tail_gradient(tail, 0, P)
        if self.exchange == "peer_compact":
            touched = self._xgrad.abs().amax(dim=1) > 0
            self._peer.all_reduce_compact_(touched)
        else:
            self._peer.all_reduce_()
        opt.begin_step()
        opt.step_range(0, P)
        opt.end_step()
        p.grad = None
        # a wait that timed out only sets a status bit and lets the stream run on (a dead peer must not wedge the GPU): read the
        # word every `peer_check_every` steps (one blocking 4-byte read), so that replicas never diverge silently for long
        self._peer_calls = getattr(self, "_peer_calls", 0) + 1
        if self._peer_calls % self.peer_check_every == 0 or self.phase_timing:
            self._peer.check_status()
        if self.phase_timing:
            self.last_exchange = {"kind": self._peer.last_kind,
                                  "bytes": self._peer.compact_bytes() if self.exchange == "peer_compact" else self._peer.last_bytes}

    def _tail_with_allreduce(self, sink):
        """Several ranks: dL/dparam must be summed before Adam.  The [P,F] table is walked in ``tail_chunks`` row ranges:
        the gradient kernel of range c is followed at once by its all-reduce (RCCL's own stream), so the collective of one
        range runs while the gradient kernels of the next ranges and the Adam kernels of the previous ones execute —
        instead of kernel, 192 MB collective, kernel in sequence."""
        opt, p = self.opt, self.model._seg_feature
        tail = opt.begin_tail(sink.rows, sink.row_grads, sink.dense)
        if tail is None:
            if self.world == 1:
                return
            p.grad = torch.zeros_like(p.data)      # nothing reached this rank's leaves: it still takes part in the sums
        bounds = row_ranges(p.shape[0], self.tail_chunks)
        works = []
        timed = self.phase_timing and self.device.type == "cuda"
        ev = (lambda: torch.cuda.Event(enable_timing=True)) if timed else None
        marks = []
        if timed:
            t_begin = ev(); t_begin.record()
        for r0, r1 in bounds:
            if tail is not None:
                opt.tail_gradient(tail, r0, r1)
            works.append(allreduce_rows_async(p.grad, r0, r1, self.world))
        if timed:
            t_grads = ev(); t_grads.record()
        opt.begin_step()
        for (r0, r1), w in zip(bounds, works):
            if timed:
                a = ev(); a.record()
            if w is not None:
                w.wait()                 # the compute stream waits for this range's collective; the host does not
            if timed:
                b = ev(); b.record()
            opt.step_range(r0, r1)
            if timed:
                c = ev(); c.record()
                marks.append((a, b, c))
        opt.end_step()
        opt.zero_grad(set_to_none=True)
        if timed:
            t_end = ev(); t_end.record()
            t_end.synchronize()
            # per row range: how long the compute stream stood still for the range's collective (0 when the collective had
            # finished under the kernels issued before it), and the range's optimiser kernel
            stalls = [round(a.elapsed_time(b), 4) for a, b, _ in marks]
            self.last_phases = {"tail_chunks": len(bounds), "gradient_kernels_ms": round(t_begin.elapsed_time(t_grads), 4),
                                "exposed_collective_ms_per_range": stalls, "exposed_collective_ms": round(sum(stalls), 4),
                                "optimizer_kernels_ms_per_range": [round(b.elapsed_time(c), 4) for _, b, c in marks],
                                "tail_ms": round(t_begin.elapsed_time(t_end), 4)}

    def _prefetch_next(self, it, after=None):
        """Software pipelining across iterations: the NEXT view's geometry pass and binning (K1, scans, key scatter, tile
        sort — atomic- and latency-bound kernels that read only frozen geometry and SH) are issued on a side stream that
        waits for this step's forward only, so they run next to this step's loss kernels (microseconds each), its backward
        and the bandwidth-bound per-Gaussian tail (and next to RCCL's all-reduce when there are several ranks).  The next
        ``render()`` waits for them and starts at the blend kernel."""
        if not self.prefetch:
            return
        if self.device.type != "cuda":
            prefetch(self.cams[self.view_index(it + 1)], self.model, self.pipe, self.bg)
            return
        if self._side is None:
            self._side = side_stream(self.device)
        # `prefetch_distance` views ahead (default 2): the chain of view it + 2 runs during step it and the first part of step
        # it + 1, so the blend of step it + 1 never waits for a chain that started only when the previous blend ended (with
        # distance 1 that chain - ~0.6 ms of kernels alone, 0.9 ms next to the loss kernels - is longer than the window
        # between two blends and the main stream idles ~0.13 ms per step at C3: profiles/r03_timeline_C3_seg.txt)
        hi = it + self.prefetch_distance
        upto = getattr(self, "_prefetched_upto", it)
        lo = upto + 1 if it <= upto <= hi else it + 1
        for t in range(lo, hi + 1):
            prefetch(self.cams[self.view_index(t)], self.model, self.pipe, self.bg, stream=self._side, after=after)
        self._prefetched_upto = hi


class PlainSegModel:
    """``GaussianModel`` as ``train_semantic.py`` sees it, getter for getter (scene/gaussian_model.py:109-138): every
    activation is re-evaluated on every call, nothing is cached."""

    def __init__(self, scene: scenes.Scene, device, class_feat=None):
        s = scene.to(device)
        self._xyz, self._scaling, self._rotation, self._opacity = s.xyz, s.log_scale, s.rot, s.opacity_logit
        self._features_dc, self._features_rest = s.features_dc, s.features_rest
        self._seg_feature = nn.Parameter(s.seg_feature.clone().requires_grad_(True))
        self.active_sh_degree = self.max_sh_degree = 3
        self.class_feat = class_feat

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))
    get_seg_feature = property(lambda s: s._seg_feature / (s._seg_feature.norm(dim=1, keepdim=True) + 1e-6))

    def get_covariance(self, scaling_modifier=1):
        return splat_to_world(self.get_xyz, self.get_scaling, scaling_modifier, self._rotation)


class PlainSegTrainer:
    """What a maintainer gets who installs the drop-in and runs ``train_semantic.py`` UNMODIFIED: the reference's iteration
    (train_semantic.py:95-208) written the way the reference writes it, on top of nothing but the two drop-in functions -
    ``render()`` and ``contrastive_loss()`` in their reference call forms.  No extension of :class:`SegTrainer` is used: a
    random view popped from a stack; the model's getters re-evaluated per call; boolean-mask gathers of the dense feature
    map (``seg_feature[:, mask][:, idx]``, the ``[F, Nv]`` temporary and its host sync); three separate losses; the dense
    zero-filled ``[F,H,W]`` gradient through the blend backward; ``torch.optim.Adam`` on ``[P,F]``; the multi-view leg
    every 10th iteration (``lambda_multiview_contras`` > 0 by default, arguments/__init__.py:114).  ``empty_cache``: also
    call ``torch.cuda.empty_cache()`` every iteration like the reference (:206).  bench.py times it as ``dropin_plain``."""

    def __init__(self, scene, cameras, device="cuda", sample_batchsize=8192, n_labels=64, lambda_sv=1e-6, lambda_mv=1e-6,
                 lambda_3d=2.5e-6, sample_mv_frames=5, use_class_feat=True, seed=0, empty_cache=False):
        import random
        self.device = torch.device(device)
        F = scene.seg_feature.shape[1]
        class_feat = None
        if use_class_feat:
            g = torch.Generator().manual_seed(seed + 17)
            class_feat = gram_schmidt(torch.rand(n_labels + 1, F, generator=g)).to(self.device)
        self.model = PlainSegModel(scene, self.device, class_feat)
        self.labels3d = scene.labels3d.to(self.device)
        self.cams = [c.to(self.device) for c in cameras]
        for i, c in enumerate(self.cams):
            if c.segmap is None:
                c.segmap = scenes.voronoi_labels(c.image_width, c.image_height, n_labels, 5000 + i, device=self.device)
                c.sorted_segmap = c.segmap
        self.pipe = PipelineParams()
        self.bg = torch.zeros(3, dtype=torch.float32, device=self.device)
        self.batch, self.lsv, self.lmv, self.l3d, self.mv_frames = sample_batchsize, lambda_sv, lambda_mv, lambda_3d, sample_mv_frames
        self.opt = torch.optim.Adam([{"params": [self.model._seg_feature], "lr": 0.025, "name": "seg_feature"}], lr=0.0, eps=1e-15)
        self.rng = random.Random(seed)
        self.stack = None
        self.empty_cache = bool(empty_cache)
        self.last_view = None

    def step(self, iteration: int):
        m, dev = self.model, self.device
        if not self.stack:
            self.stack = list(range(len(self.cams)))
        vi = self.stack.pop(self.rng.randint(0, len(self.stack) - 1))
        self.last_view = vi
        cam = self.cams[vi]
        pkg = render(cam, m, self.pipe, self.bg)
        seg_feature, visibility_filter = pkg["seg_feature"], pkg["visibility_filter"]
        loss = 0
        segmaps = [cam.segmap, cam.sorted_segmap] if m.class_feat is not None else [cam.segmap]
        for k, gt in enumerate(segmaps):
            valid = gt > 0
            if valid.sum() > 0:
                feats, labels = seg_feature[:, valid], gt[valid]
                idx = torch.randint(0, len(labels), size=(self.batch,), device=dev)
                loss = loss + contrastive_loss(feats[:, idx].T, labels[idx], predef_u_list=m.class_feat if k == 1 else None) \
                    * self.lsv * (1 if k == 1 else 0.5)
        if self.lmv > 0 and iteration % 10 == 0 and len(self.cams) > self.mv_frames:      # (the reference's np.random.randint needs it too)
            first = self.rng.randint(0, len(self.cams) - self.mv_frames - 1)
            views = self.cams[first:first + self.mv_frames]
            maps = torch.stack([render(v, m, self.pipe, self.bg)["seg_feature"] for v in views], dim=0)
            labs = torch.stack([v.sorted_segmap for v in views], dim=0)
            valid = labs > 0
            feats, labels = maps.permute(1, 0, 2, 3)[:, valid], labs[valid]
            idx = torch.randint(0, len(labels), size=(self.batch,), device=dev)
            loss = loss + contrastive_loss(feats[:, idx].T, labels[idx], predef_u_list=m.class_feat) * self.lmv
        vis_feat, vis_lab = m.get_seg_feature[visibility_filter], self.labels3d[visibility_filter]
        if self.l3d > 0:
            valid = vis_lab > 0
            if valid.sum() > 0:
                feats, labels = vis_feat[valid], vis_lab[valid]
                idx = torch.randint(0, len(labels), size=(self.batch,), device=dev)
                loss = loss + contrastive_loss(feats[idx], labels[idx], predef_u_list=m.class_feat) * self.l3d
        loss.backward()
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        if self.empty_cache:
            torch.cuda.empty_cache()
        return loss.detach()


class RgbGaussianModel:
    """Trainable 2DGS model for the ``train.py``-style step (scene/gaussian_model.py:109-138,206-253): six
    parameter groups with the reference's learning rates; densification is out of scope."""

    def __init__(self, scene: scenes.Scene, device):
        s = scene.to(device)
        mk = lambda t: nn.Parameter(t.clone().requires_grad_(True))
        self._xyz, self._scaling, self._rotation = mk(s.xyz), mk(s.log_scale), mk(s.rot)
        self._opacity, self._features_dc, self._features_rest = mk(s.opacity_logit), mk(s.features_dc), mk(s.features_rest)
        self.active_sh_degree = 3
        self.max_sh_degree = 3
        self._leaves = None      # optim.GaussianAdam.begin(): the activated tensors as leaves for the duration of a step

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: s._leaves["scaling"] if s._leaves else torch.exp(s._scaling))
    get_rotation = property(lambda s: s._leaves["rotation"] if s._leaves else torch.nn.functional.normalize(s._rotation))
    get_opacity = property(lambda s: s._leaves["opacity"] if s._leaves else torch.sigmoid(s._opacity))
    get_features = property(lambda s: s._leaves["shs"] if s._leaves else torch.cat((s._features_dc, s._features_rest), dim=1))
    get_seg_feature = property(lambda s: None)

    def get_covariance(self, scaling_modifier=1):
        # (with the fused optimiser's leaves active the gradient must reach leaves["rotation"] - the normalised quaternion,
        # which rotation_matrices re-normalises harmlessly - not the raw parameter, whose .grad that optimiser never reads)
        return splat_to_world(self.get_xyz, self.get_scaling, scaling_modifier, self._leaves["rotation"] if self._leaves else self._rotation)

    def param_groups(self):
        return [{"params": [self._xyz], "lr": 0.00016, "name": "xyz"},
                {"params": [self._features_dc], "lr": 0.0025, "name": "f_dc"},
                {"params": [self._features_rest], "lr": 0.0025 / 20.0, "name": "f_rest"},
                {"params": [self._opacity], "lr": 0.05, "name": "opacity"},
                {"params": [self._scaling], "lr": 0.005, "name": "scaling"},
                {"params": [self._rotation], "lr": 0.001, "name": "rotation"}]


class RgbTrainer:
    """Counterpart of train.py:57-156 without densification: render -> (1-l)*L1 + l*(1-SSIM) + lambda_dist*dist +
    lambda_normal*(1 - <rend_normal, surf_normal>) -> backward (full geometry gradients) -> Adam."""

    def __init__(self, scene, cameras, targets, device="cuda", lambda_dssim=0.2, lambda_normal=0.05, lambda_dist=0.0,
                 rank=0, world=1, densify=None, scene_extent=None, spatial_sort=True, fused_update=None):
        """``spatial_sort``: store the Gaussians in Z-order of their centres (a row permutation, once, here; rows that the
        density control appends later go to the end).  ``densify``: None (off) or a dict overriding the reference's schedule (arguments/__init__.py:106-125):
        ``from_iter=500, until_iter=15000, interval=100, opacity_reset_interval=3000, grad_threshold=0.0002,
        opacity_cull=0.05, percent_dense=0.01``; ``scene_extent`` = the reference's ``cameras_extent``."""
        from .losses import l1_loss, photometric_loss, ssim, train_loss
        self.l1, self.ssim, self.photometric, self.train_loss = l1_loss, ssim, photometric_loss, train_loss
        self.device = torch.device(device)
        self.order = None
        if spatial_sort:
            self.order = scenes.morton_order(scene.xyz)
            scene = scenes.spatially_sorted(scene, self.order)
        self.model = RgbGaussianModel(scene, self.device)
        self.cams = [c.to(self.device) for c in cameras]
        self.targets = [t.to(self.device) for t in targets]
        self.pipe = PipelineParams()
        self.bg = torch.zeros(3, dtype=torch.float32, device=self.device)
        self.ld, self.ln, self.ldist = lambda_dssim, lambda_normal, lambda_dist
        self.fused_loss = self.device.type == "cuda"        # tests switch it off to compare with the composed form
        self.c_tail = os.environ.get("ISR_C_TAIL", "1") != "0"      # the iteration behind render() as one C entry (_c_tail_ok)
        self.rank, self.world = rank, world
        # chain rule of the getters + Adam on the six groups + the next activations in one kernel (optim.GaussianAdam);
        # the density control edits optimiser rows through torch's state dict, so it keeps torch.optim.Adam
        if fused_update is None:
            fused_update = self.device.type == "cuda" and densify is None
        if fused_update and densify is not None:
            raise ValueError("fused_update does not support the density control")
        self.fused_update = bool(fused_update)
        if self.fused_update:
            from .optim import GaussianAdam
            self.opt = GaussianAdam(self.model, {g["name"]: g["lr"] for g in self.model.param_groups()}, eps=1e-15)
        else:
            self.opt = torch.optim.Adam(self.model.param_groups(), lr=0.0, eps=1e-15, fused=self.device.type == "cuda")
        self.densify_cfg = None
        if densify is not None:
            from .densify import Densifier
            self.densify_cfg = dict(from_iter=500, until_iter=15_000, interval=100, opacity_reset_interval=3000,
                                    grad_threshold=0.0002, opacity_cull=0.05, percent_dense=0.01)
            self.densify_cfg.update(densify)
            self.densifier = Densifier(self.model, self.opt, self.densify_cfg["percent_dense"])
            if scene_extent is None:      # radius of the camera ring around its centroid, x1.1 (scene/dataset_readers.py)
                centers = torch.stack([c.camera_center for c in self.cams])
                scene_extent = float((centers - centers.mean(dim=0)).norm(dim=1).max()) * 1.1
            self.scene_extent = float(scene_extent)

    def _density_control(self, iteration: int, pkg) -> None:
        """train.py:138-151, between backward and the optimiser step; with several ranks the statistics are summed /
        maxed first and the split draws from an identically seeded stream, so every replica edits the same rows."""
        cfg, d = self.densify_cfg, self.densifier
        if iteration >= cfg["until_iter"]:
            return
        d.accumulate(pkg["viewspace_points"].grad, pkg["visibility_filter"], pkg["radii"])
        if iteration > cfg["from_iter"] and iteration % cfg["interval"] == 0:
            if self.world > 1:
                dist.all_reduce(d.xyz_gradient_accum, op=dist.ReduceOp.SUM)
                dist.all_reduce(d.denom, op=dist.ReduceOp.SUM)
                dist.all_reduce(d.max_radii2D, op=dist.ReduceOp.MAX)
                torch.manual_seed(977 + iteration)
            size_threshold = 20 if iteration > cfg["opacity_reset_interval"] else None
            d.densify_and_prune(cfg["grad_threshold"], cfg["opacity_cull"], self.scene_extent, size_threshold)
        if iteration % cfg["opacity_reset_interval"] == 0:
            d.reset_opacity()

    def _loss(self, pkg, vi):
        image, gt = pkg["render"], self.targets[vi]
        if self.fused_loss:
            # photometric term + both regularisers: three launches forward, one backward (losses.train_loss)
            return self.train_loss(image, gt, self.ld, pkg["rend_normal"], pkg["surf_normal"], self.ln,
                                   pkg["rend_dist"] if self.ldist != 0.0 else None, self.ldist)
        loss = self.photometric(image, gt, self.ld)       # (1 - l) L1 + l (1 - SSIM), one pair of kernels on the GPU
        if self.ldist != 0.0:
            loss = loss + self.ldist * pkg["rend_dist"].mean()
        normal_error = (1 - (pkg["rend_normal"] * pkg["surf_normal"]).sum(dim=0))[None]
        return loss + self.ln * normal_error.mean()

    def view_index(self, it):
        return view_for(it, self.rank, self.world, len(self.cams))

    def step(self, it: int):
        from .rasterizer import BinningOverflow
        try:
            return self._step_once(it)
        except BinningOverflow:
            # the view outgrew its async binning estimate (the geometry trains): the estimate is corrected, nothing has
            # reached the parameters (the check precedes the backward kernels) - run the iteration again
            self.opt.zero_grad(set_to_none=True)
            return self._step_once(it)

    # -- everything behind render() through ONE C entry (isr_rgb_step_tail) instead of the autograd graph ---------------------
    def _c_tail_ok(self, pkg) -> bool:
        """One rank, the fused loss and the fused optimiser: the loss, its backward through render()'s derived maps, the
        rasterizer's backward and the six-group Adam step are then ONE host call (the same launches in the order of the autograd
        graph: bit-identical parameters) instead of three autograd Functions, a graph and a backward pass of the engine - the C2 step
        is host-paced otherwise (0.78 ms of Python for 0.62 ms of kernels).  ISR_C_TAIL=0: always the autograd path."""
        if not (self.c_tail and self.fused_update and self.fused_loss and self.world == 1 and self.device.type == "cuda"):
            return False
        img = pkg["render"]
        n1 = getattr(img, "grad_fn", None)
        n2 = getattr(pkg["rend_normal"], "grad_fn", None)
        return (n1 is not None and hasattr(n1, "num_rendered") and n2 is not None and hasattr(n2, "ratio")
                and self.opt.leaves is not None and getattr(self.model, "_seg_feature", None) is None)

    def _c_tail(self, pkg, vi):
        import ctypes
        from . import _hot, arena, rasterizer as _rz
        from ._lib import GRAD_GEOMETRY, check, lib
        from .optim import GROUPS, _ATTR, _table
        L = lib()
        dev, opt, m = self.device, self.opt, self.model
        image, gt = pkg["render"], self.targets[vi]
        n1, n2 = image.grad_fn, pkg["rend_normal"].grad_fn
        (_, means3D, scales, rotations, _, radii, _, sh, geom, binning, img) = n1.saved_tensors
        am, vm, rays_d, rays_o, surf = n2.saved_tensors
        rs, R, mode = n1.raster_settings, int(n1.num_rendered), int(n1.mode)
        W, H = int(rs.image_width), int(rs.image_height)
        P = means3D.shape[0]
        M = sh.shape[1]
        _rz._verify_pending(geom.data_ptr())        # (async binning: BinningOverflow before anything reaches the parameters)
        use_d = self.ldist != 0.0
        f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        ar = lambda *shape: arena.empty(shape, torch.float32, dev)
        gtc = gt.detach().contiguous().float()
        rn, sn = pkg["rend_normal"].detach(), pkg["surf_normal"].detach()
        rd = pkg["rend_dist"].detach() if use_d else None
        loss5, dmaps = f32(5), f32(3, 3, H, W)
        nb = L.iso_train_loss_scratch_bytes(3, H, W)
        lscratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        d_img, d_rn, d_sn = f32(3, H, W), f32(3, H, W), f32(3, H, W)
        d_rd = f32(*rd.shape) if use_d else None
        pscratch, d_all = f32(6, H, W), f32(7, H, W)
        g2, gn, go, gc, g3, gtm, gsh, gs, gr = ar(P, 3), ar(P, 3), ar(P, 1), ar(P, 3), ar(P, 3), ar(P, 9), ar(P, M, 3), ar(P, 2), ar(P, 4)
        bscratch = _rz._workspace(lambda c: L.isr_backward_scratch_bytes(c, 0, GRAD_GEOMETRY), R, dev)
        acts = (f32(P, M, 3), f32(P, 1), f32(P, 2), f32(P, 4))
        ps = opt._params()
        lr = (ctypes.c_double * 6)(*[float(g["lr"]) for g in opt.param_groups])
        one = getattr(self, "_one_c", None)
        if one is None or one.device != dev:
            one = self._one_c = torch.ones(1, dtype=torch.float32, device=dev)
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        c32 = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()
        opt.step_count += 1
        with _hot.on_device(dev):
            check(L.isr_rgb_step_tail(
                P, int(rs.sh_degree), M, W, H, mode, R, p(image.detach()), p(gtc), p(am), p(rn), p(sn), p(rd), p(surf),
                float(self.ld), float(self.ln), float(self.ldist), float(n2.ratio), p(rays_d), p(rays_o),
                p(c32(rs.bg)), p(means3D), p(sh), p(scales), float(rs.scale_modifier), p(rotations), p(c32(rs.viewmatrix)),
                p(c32(rs.projmatrix)), p(c32(rs.campos)), float(rs.tanfovx), float(rs.tanfovy), p(radii), p(geom), p(binning), p(img),
                _table([t.data for t in ps]), _table([opt.exp_avg[k] for k in GROUPS]), _table([opt.exp_avg_sq[k] for k in GROUPS]),
                lr, float(opt.betas[0]), float(opt.betas[1]), opt.eps, max(1, opt.step_count), p(acts[0]), p(acts[1]), p(acts[2]),
                p(acts[3]), p(loss5), p(dmaps), p(lscratch), nb, p(d_img), p(d_rn), p(d_sn), p(d_rd), p(pscratch), p(d_all),
                p(g2), p(gn), p(go), p(gc), p(g3), p(gtm), p(gsh), p(gs), p(gr), p(bscratch), bscratch.numel(), p(one),
                _hot.stream_ptr(dev)), "isr_rgb_step_tail")
        for prm in ps:
            torch.autograd.graph.increment_version(prm)         # the kernels wrote through raw pointers
        opt._acts = (tuple(prm._version for prm in ps), acts)
        opt.leaves = None
        return loss5[0]

    def _step_once(self, it: int):
        vi = self.view_index(it)
        if self.fused_update:
            self.model._leaves = self.opt.begin()
            try:
                pkg = render(self.cams[vi], self.model, self.pipe, self.bg)
                if self._c_tail_ok(pkg):
                    return self._c_tail(pkg, vi), pkg
                loss = self._loss(pkg, vi)
                loss.backward(_unit_grad(self, loss))
            finally:
                self.model._leaves = None
            # one flat collective for all six groups' gradients (they are final only after the per-Gaussian backward pass)
            lv = self.opt.leaves
            self.opt.step(allreduce_bucket(self.opt.leaf_grads(), self.world,
                                           like=[lv["xyz"], lv["shs"], lv["opacity"], lv["scaling"], lv["rotation"]]))
            self.opt.zero_grad(set_to_none=True)
            return loss.detach(), pkg
        pkg = render(self.cams[vi], self.model, self.pipe, self.bg)
        loss = self._loss(pkg, vi)
        loss.backward()
        allreduce_grads([p for gr in self.opt.param_groups for p in gr["params"]], self.world)
        if self.densify_cfg is not None:
            self._density_control(it + 1, pkg)          # the reference counts iterations from 1
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        return loss.detach(), pkg
