#!/usr/bin/env python
"""Benchmark of the hot path named by BASELINE.json.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3|C5|C2|C1] [--step seg|rgb] [--mode fast|exact|fast_reflists]

* ``--step seg`` (default, configs with a feature channel: C3 = the headline, C5): the train_semantic.py step - render
  forward, two single-view contrastive losses on 8 192 sampled pixels each, the 3-D contrastive loss, backward, Adam on
  the [P,F] feature.  ``--step rgb`` (C2 = BASELINE config 2, also C1 / C3): the train.py step - render, L1 + SSIM +
  normal consistency, full geometry backward, Adam on the six parameter groups.
* ``--mode fast_reflists`` (default, the headline, = the library's shipped default): FAST arithmetic in the per-pixel loops
  (decisions are EXACT's by construction, images within 1e-4, gradients within 1e-3) on the REFERENCE's tile rectangles: radii,
  tiles_touched, point_list, ranges and num_rendered bit-identical to the reference's.  At one GPU the same line also carries
  ``sub_records`` for ``exact`` (op-for-op IEEE, all integer state and images bit-identical to the CPU oracle) and ``fast``
  (opt-in: tile lists that hold a splat only where its alpha >= 1/255 box reaches - the kernels walk the same (block, splat)
  pairs, so every output except the last bits of the distortion channel is the same; ~2 % faster), timed the same way.
* ``--gpus N`` > 1 without a launcher re-executes itself under ``torch.distributed.run`` (one rank per GPU, RCCL); under
  a launcher (WORLD_SIZE set) it is a rank.  Every rank renders a different view per step (weak scaling); the parameter
  gradients are summed across ranks.  Rank 0 prints ONE JSON line.
* Output: the LAST line of stdout is a compact record (``compact_record``: <= 6 KB - the contract keys, the numeric contract keys
  of ``config``, the dominant kernel's ``roofline`` with the numbers its fractions are recomputable from, ``cpu_baseline``); the
  full record with ``sub_records``, per-kernel tables and notes goes to ``bench_details.json`` (repo root, and ``gpurun_out/`` when
  that exists) and, as one line prefixed ``[bench details]``, to stderr.  (Round 5 printed the full record - 24.8 KB - as the
  final line, which the driver's parser lost.)
"""
import argparse
import ctypes
import json
import math
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEFAULT_MODE = "fast_reflists"  # the library's shipped default: FAST arithmetic on the reference's tile lists (integer state bit-exact)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E (MI355X_MICROARCH.md); measured copy peak ~6290 GB/s
VALU_PEAK_TFLOPS = 157.3       # fp32 vector peak


SIMDS, CLOCK_GHZ = 1024, 2.4      # MI355X: 256 CUs x 4 SIMDs
# Vector-instruction issue, MEASURED on this part (tools/micro/valu_issue.hip, profiles/r05_valu_issue.txt): wave-instructions per
# second the whole chip sustains at 4-8 waves per SIMD.  Plain single-issue fp32 (v_fma / v_mul / v_add / v_mov): 820-850 G/s
# (= 2.9 cycles per instruction and SIMD at the nominal 2.4 GHz; the in-kernel cycle counter says 2.1 at the 2.0-2.3 GHz the part
# really runs at under that load - the guide's "2 cycles", MI355X_MICROARCH.md:52-54,430 - round 4 assumed 4).  Packed
# (v_pk_fma_f32), SGPR-writing compares, v_cndmask with an SGPR mask, DPP, v_readlane: 500-570 G/s (x1.5); v_rcp / v_exp /
# v_permlane32_swap: 290-300 G/s (x2.9); ds_read_b128: 150 G/s for any address pattern (4 cycles per CU).
ISSUE_PEAK_PLAIN_G = 850.0
# the blend kernel's walk, counted by class in its ISA: ~77 plain-equivalent issue slots for 66 vector instructions
ISSUE_MIX_WEIGHT = 77.0 / 66.0


def csrc_tree_hash():
    """sha256 over the library's sources: profiles/roofline_traffic.json records the tree its PMC counters were taken on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "instascene_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def _self_launch(n):
    """``python bench.py --gpus N`` with no launcher: become ``torch.distributed.run`` with N ranks on this node."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("OMP_NUM_THREADS", "8")
    os.execv(sys.executable, cmd)


def byte_model(P, V, R, N, F, tiles):
    """SURVEY.md 8(d) / DESIGN.md section 3: algorithmic HBM bytes per view, per kernel of this design."""
    return {
        "k_render_fwd": R * (64 + 12 + 4 * F) + N * (40 + 20 + 4 * F),      # K8: staged records/colours/features + per-pixel outputs
        "k_render_bwd_dense": N * (40 + 20 + 4 * F) + R * (64 + 12 + 4 * F) + R * 2 * 4 * (16 + F),   # K9, dense upstream gradient
        "k_preprocess": 12 * P + 220 * V + 92 * V + 20 * P + 32 * V,         # xyz; scale/rot/opacity/SH in; record, rect, counts, radii, cull out
        "k_scatter": 16 * V + 8 * R,
        "k_tile_sort": 12 * R,
        "k_pack_hits": R * (4 + 32 + 4) + R // 2,                           # id + K1's bounds in; box4 + the per-chunk hit masks out
        "k_feature_rows_step": 8 * P + 7 * 4 * F * P,                        # mask/offsets + x, m, v in; x, m, v, z out (+ flagged rows)
        "k_preprocess_bwd": 340 * V + 252 * P,
        "pp_maps": (28 + 44) * N,
        "pp_surf_normal": (20 + 12) * N,
    }


def profile_summary(L):
    buf = ctypes.create_string_buffer(1 << 16)
    L.isr_profile_summary(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, tot = line.split()
        out[name] = (int(cnt), float(tot))
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, step, n_samples, seconds_budget=25.0):
    """The CPU oracle (C++/OpenMP restatement of the reference kernels; the reference has no CPU path of its own) timed on
    this box's host cores on a bounded sample of the same per-view work: forward + the reference's backward of one view, at
    the largest scale (1, 1/2 or 1/4 in each image dimension, P scaled alike) whose estimated time fits the budget."""
    import numpy as np
    import oracle
    from instascene_amd import scenes
    flags = oracle.use_native_build()          # g++ -O3 -march=native on this box; same bits as the tests' build
    F = cfg["F"] if step == "seg" else 0

    def build(scale):
        P = int(cfg["P"] * scale * scale)
        W, H = int(cfg["W"] * scale), int(cfg["H"] * scale)
        sc = scenes.synthetic_scene(P, F, scenes.SEED_BASE + cfg["index"], cfg["mu_s"] - math.log(scale))
        cam = scenes.ring_cameras(64, W, H)[0]
        a = {k: (None if v is None else v.numpy()) for k, v in scenes.activated_inputs(sc).items()}
        return P, W, H, cam, a

    def one(P, W, H, cam, a):
        st = oracle.forward(a["means3D"], a["opacities"], cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(),
                            cam.camera_center.numpy(), np.zeros(3, np.float32), W, H, math.tan(cam.FoVx / 2),
                            math.tan(cam.FoVy / 2), scales=a["scales"], rotations=a["rotations"], shs=a["shs"],
                            extra=a["extra"], sh_degree=3)
        if step == "seg":
            # the upstream gradient of this step: dL/dfeature on the sampled pixels only, nothing on colour / aux maps
            dE = np.zeros_like(st["extra"]).reshape(F, -1)
            pick = np.random.RandomState(0).randint(0, W * H, max(1, int(n_samples * W * H / (cfg["W"] * cfg["H"]))))
            dE[:, pick] = 1.0
            oracle.backward(st, np.zeros_like(st["color"]), np.zeros_like(st["others"]), dE.reshape(st["extra"].shape))
        else:
            oracle.backward(st, np.ones_like(st["color"]), np.ones_like(st["others"]), None)
        return st["R"]

    small = build(0.25)     # splats enlarged by 1/scale in world units: same pixel footprint and depth complexity
    one(*small)
    t0 = time.time()
    one(*small)
    t_small = time.time() - t0
    scale = 0.25
    for s_try in (1.0, 0.5):
        if t_small * (s_try / 0.25) ** 2 * 1.3 <= seconds_budget:
            scale = s_try
            break
    args = small if scale == 0.25 else build(scale)
    t0 = time.time()
    n, R = 0, 0
    while True:
        R = one(*args)
        n += 1
        if time.time() - t0 > 10.0 or n >= 8:
            break
    dt = (time.time() - t0) / n
    P, W, H = args[0], args[1], args[2]
    frac = scale * scale
    what = ("forward + the reference's backward walk over all pixels of one view with dL/dfeature on the sampled pixels only "
            "(the reference does not exploit that sparsity, the GPU path does; the three losses and Adam are not in the sample)"
            if step == "seg" else "forward + dense full backward (colour + aux maps) of one view (losses and Adam not in the sample)")
    return {"value": frac / dt, "unit": "views/s", "cores": oracle.num_threads(), "cpu": cpu_model(), "kind": "port",
            "build": "g++ " + flags,
            "sample": f"oracle (g++ {flags.split(' -std')[0]}) {what}; {cfg['name']} at scale {scale:g} (P={P}, {W}x{H}, F={F}, R={R}): {n} view(s) in "
                      f"{dt * n:.1f} s" + ("" if scale == 1.0 else f"; value = measured {1.0 / dt:.3f} views/s x {frac:g} "
                                           "(work scales with P and pixels)")}


def time_allreduce(numel, dev, world, reps=5):
    import torch
    import torch.distributed as dist
    if world <= 1:
        return None
    t = torch.zeros(numel, dtype=torch.float32, device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(t)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def run(args, mode, rank, world, dev, detail, repeats=1):
    """One timed measurement in ``mode``: W warm-up steps, then exactly K steps between barrier + synchronize on both
    sides.  Returns the record (rank 0: with the kernel detail when ``detail``)."""
    import torch
    import torch.distributed as dist
    from instascene_amd import scenes, rasterizer
    from instascene_amd._lib import lib
    from instascene_amd.harness import RgbTrainer, SegTrainer
    from instascene_amd.render import render

    feature_only = mode.endswith("+feature_only")      # opt-in sub-record: the blend kernel skips colour / aux maps / tracer
    mode = mode.split("+")[0]
    rasterizer.set_mode(mode)
    rasterizer.set_tracer(bool(args.tracer))
    rasterizer.set_async_binning(bool(args.async_binning))
    rasterizer.set_view_cache(args.view_cache_gb)
    scene, cams, cfg = scenes.config_scene(args.config)
    cfg["name"] = args.config
    if getattr(args, "feat_dim", None) and args.step in ("seg", "plain") and int(args.feat_dim) != cfg["F"]:
        # the same scene with another feature width (the reference's default seg_feat_dim is 16)
        cfg = dict(cfg, F=int(args.feat_dim))
        scene, cams = scenes.synthetic_scene(cfg["P"], cfg["F"], scenes.SEED_BASE + scenes.CONFIGS[args.config]["index"], cfg["mu_s"]), cams
    n_views = 16
    plain = args.step == "plain"
    if plain:
        # the reference's loop on the drop-in functions alone, library defaults (no async binning, tracer on): PlainSegTrainer
        from instascene_amd.harness import PlainSegTrainer
        rasterizer.set_async_binning(False)
        rasterizer.set_tracer(True)
        trainer = PlainSegTrainer(scene, cams[:n_views], device=dev, sample_batchsize=8192, use_class_feat=True,
                                  empty_cache=bool(getattr(args, "empty_cache", False)))
        view_index = lambda it: trainer.last_view
    elif args.step == "seg":
        trainer = SegTrainer(scene, cams[:n_views], device=dev, sample_batchsize=int(getattr(args, "sample_batchsize", 8192)),
                             use_class_feat=True, rank=rank, world=world,
                             spatial_sort=bool(args.spatial_sort), fused_sampling=bool(args.fused_sampling),
                             multiview=bool(getattr(args, "multiview", False)))
        trainer.split_tail = bool(args.split_tail)
        if args.sharded_tail is not None:
            trainer.sharded_tail = bool(args.sharded_tail)
        if getattr(args, "exchange", None):
            trainer.exchange = args.exchange
        trainer.pipe.lazy_maps = bool(args.lazy_maps)
        trainer.pipe.feature_only_forward = feature_only
        trainer.warm_view_caches()       # per-view constants (ray tables, visible pools, instance counts): setup
        trainer.prime(steps=n_views)     # code objects, allocator / arena pools, side stream, clocks: one pass over the views whose effect is undone
        view_index = trainer.view_index
    else:
        scene.seg_feature = None
        trainer = RgbTrainer(scene, cams[:n_views], [torch.zeros(3, 8, 8)] * n_views, device=dev, rank=rank, world=world,
                             spatial_sort=bool(args.spatial_sort))
        with torch.no_grad():            # targets: the initial renders plus noise (setup)
            g = torch.Generator(device=dev).manual_seed(5)
            trainer.targets = [(render(c, trainer.model, trainer.pipe, trainer.bg)["render"]
                                + 0.05 * torch.randn(3, cfg["H"], cfg["W"], device=dev, generator=g)).clamp(0, 1)
                               for c in trainer.cams]
        view_index = trainer.view_index
    L = lib()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import contextlib
    step_scope = trainer.stream_scope if (hasattr(trainer, "stream_scope") and os.environ.get("ISR_BENCH_SCOPE", "1") == "1") \
        else contextlib.nullcontext
    with step_scope():
        for it in range(args.warmup):
            trainer.step(it)
    sync()
    from instascene_amd import arena as _arena
    dbg0 = (rasterizer.PREFETCH_HITS, torch.cuda.memory_stats().get("num_device_alloc", 0), len(rasterizer._R_ESTIMATE),
            _arena.STATS["new_blocks"])
    dt, it_next = None, args.warmup
    for rep in range(repeats):       # the headline: exactly one block of K steps; sub-records: the better of two blocks
        L.isr_profile_enable(2)      # HIP events around the forward blend kernel only inside the timed region
        t0 = time.perf_counter()
        with step_scope():           # the trainer's own stream for the whole block (SegTrainer.stream_scope)
            for it in range(it_next, it_next + args.steps):
                trainer.step(it)
        sync()
        dt_rep = time.perf_counter() - t0
        if dt is None or dt_rep < dt:
            dt = dt_rep
            prof_dom = profile_summary(L)
        L.isr_profile_enable(0)
        it_next += args.steps
    if os.environ.get("ISR_BENCH_DEBUG"):
        print(f"[bench debug] mode={mode} prefetch hits {rasterizer.PREFETCH_HITS - dbg0[0]} / {args.steps} steps, device allocs "
              f"{torch.cuda.memory_stats().get('num_device_alloc', 0) - dbg0[1]}, estimates {dbg0[2]} -> {len(rasterizer._R_ESTIMATE)}, "
              f"pending {len(rasterizer._PENDING)}, reserved {torch.cuda.memory_reserved() / 2**30:.2f} GB, arena new blocks "
              f"{_arena.STATS['new_blocks'] - dbg0[3]} ({_arena.reserved_bytes() / 2**30:.2f} GB)", file=sys.stderr)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    rec = {"value": round(world * args.steps / dt, 3), "ms_per_step": round(1e3 * dt / args.steps, 4), "steps": args.steps,
           "arithmetic_mode": mode + ("+feature_only" if feature_only else ""),
           "sharded_tail": bool(getattr(trainer, "sharded_tail", False))}
    # (steps with a multi-view leg every 10th iteration: the detail pass covers exactly one such iteration)
    extra_steps = 10 if (plain or getattr(args, "multiview", False)) else min(5, args.steps)
    it0 = it_next
    if detail:
        # every kernel of the library, over a few extra (untimed) steps, HIP events on the launch stream (all ranks step:
        # a step holds collectives)
        L.isr_profile_enable(1)
        for it in range(it0, it0 + extra_steps):
            trainer.step(it)
        sync()
        prof_all = profile_summary(L)
        L.isr_profile_enable(0)
    if rank == 0 and detail:
        # workload statistics + work counters of the blend kernel on the last timed view (one extra, untimed render)
        counters = torch.zeros(16, dtype=torch.int64, device=dev)
        with torch.no_grad():
            was = rasterizer._CONFIG["async_binning"]
            rasterizer.set_async_binning(False)             # exact instance count for the byte model
            if mode != "exact":
                L.isr_forward_set_counters(ctypes.c_void_p(counters.data_ptr()))
            trainer.pipe.feature_only_forward = False
            trainer.model._seg_cache = None
            pkg = render(trainer.cams[view_index(it0 - 1)], trainer.model, trainer.pipe, trainer.bg)
            V = int((pkg["radii"] > 0).sum().item())
            R = int(rasterizer.LAST_NUM_RENDERED)
            rasterizer.set_async_binning(was)
        cull_tests, pairs_eval, pairs_blend, lane_pairs, pairs_mergeable, sub_blocks, pairs_exact, band_violations = (int(v) for v in counters.tolist()[:8])
        sublists = [int(v) for v in counters.tolist()[8:16]]
        P, N, F = cfg["P"], cfg["W"] * cfg["H"], (cfg["F"] if args.step in ("seg", "plain") else 0)
        tiles = ((cfg["W"] + 15) // 16) * ((cfg["H"] + 15) // 16)
        bm = byte_model(P, V, R, N, F, tiles)
        kern = {}
        for k, (cnt, tot) in sorted(prof_all.items()):
            ms = tot / cnt
            key = "k_render_bwd_dense" if (k == "k_render_bwd" and args.step in ("rgb", "plain")) else k
            e = {"ms_per_launch": round(ms, 4), "launches_per_view": round(cnt / float(extra_steps), 2)}
            if key in bm:
                # a launch of a blend kernel covers one 32-channel chunk of one view; every other kernel one view
                # a view's bytes over the launches it took (blend kernels: one per 32-channel chunk, or per 64 in the wide pass)
                per_launch = bm[key] / (max(1, int(round(cnt / float(extra_steps)))) if key in ("k_render_fwd", "k_render_bwd_dense") else 1)
                e["algorithmic_bytes"] = int(per_launch)
                e["GB/s"] = round(per_launch / (ms * 1e-3) / 1e9, 1)
                e["frac_hbm"] = round(per_launch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            kern[k] = e
        dom = "k_render_fwd" if args.step == "seg" and not getattr(args, "multiview", False) else \
            max(kern, key=lambda k: kern[k]["ms_per_launch"] * kern[k]["launches_per_view"])
        dom_ms = kern[dom]["ms_per_launch"]
        timing = "HIP events on the launch stream over %d extra untimed steps" % extra_steps
        if dom in prof_dom:              # the dominant forward kernel: measured over the timed region itself
            dom_ms = prof_dom[dom][1] / prof_dom[dom][0]
            kern[dom]["ms_per_launch"] = round(dom_ms, 4)
            timing = "HIP events on the launch stream: %s over the timed region, the other kernels over %d extra untimed steps" % (dom, extra_steps)
        dom_key = "k_render_bwd_dense" if (dom == "k_render_bwd" and args.step in ("rgb", "plain")) else dom
        launches = max(1, int(round(kern[dom]["launches_per_view"]))) if dom_key in ("k_render_fwd", "k_render_bwd_dense") else 1
        dom_bytes = bm.get(dom_key, 0) / launches
        gbs = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic, tsrc, issue, stale = None, None, None, None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # (the counters are taken in the default mode; the FAST modes run the same kernels on lists of slightly different length)
                tmode = mode if (args.config + ":" + args.step + ":" + mode) in tj else DEFAULT_MODE
                traffic = tj.get(args.config + ":" + args.step + ":" + tmode, {}).get(dom)
                tsrc = tj.get("source")
                issue = tj.get(args.config + ":" + args.step + ":" + tmode + ":" + dom + ":issue")
                stale = tj.get("csrc_tree_hash") != csrc_tree_hash()      # the counters were taken on another tree
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": dom, "traffic_stale": stale, "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": tsrc,
                "traffic_over_algorithmic_bytes": (round(traffic / dom_bytes, 3) if (traffic and dom_bytes) else None),
                "avg_launch_ms": round(dom_ms, 4), "launches_per_view": launches,
                "launches_per_step": kern[dom]["launches_per_view"],
                "hbm": {"bytes": int(dom_bytes), "GB/s": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 5),
                        "frac_of_measured_copy_peak_6290": round(gbs / 6290.0, 5)},
                "note": "frac = SURVEY 8(d): max(algorithmic bytes / 8 TB/s, model flops / 157.3 TF) of the dominant kernel.  The blend kernel "
                        "is bound by neither but by vector-instruction issue (`issue`): one wave per 8x8 block walks its hit list "
                        "serially with a third of the lanes of a blending evaluation blending (`valu`); DESIGN.md section 3",
                "timing": timing, "kernels": kern,
                "workload": {"P": P, "V": V, "R": R, "N": N, "F": F, "tiles": tiles}}
        if dom == "k_render_fwd" and pairs_eval:
            per_pair, per_contrib = 40.0, 2.0 * (3 + 7 + F)
            flops_eval = per_pair * 64 * pairs_eval + per_contrib * lane_pairs
            flops_model = per_pair * 256.0 * R + per_contrib * lane_pairs
            tf = flops_eval / (dom_ms * launches * 1e-3) / 1e12
            roof["valu"] = {
                "wave_splat_cull_tests": cull_tests, "wave_splat_pairs_evaluated": pairs_eval,
                "wave_splat_pairs_blending": pairs_blend,
                # guard bands (csrc/isr_fast_pair.hpp): evaluations that took EXACT's instruction sequence, and pairs OUTSIDE the bands
                # whose decision differs from EXACT's (the band's bound, checked on the device: must be 0)
                "wave_splat_pairs_on_the_exact_path": pairs_exact, "pairs_outside_the_guard_bands_deciding_unlike_exact": band_violations,
                "pixel_splat_pairs_evaluated": 64 * pairs_eval,
                "pixel_splat_pairs_contributing": lane_pairs,
                "lane_utilisation_of_blending_pairs": round(lane_pairs / max(1, 64 * pairs_blend), 4),
                "blending_4x4_sub_blocks_per_blending_pair": round(sub_blocks / max(1, pairs_blend), 4),
                "lane_utilisation_at_4x4_granularity": round(lane_pairs / max(1, 16 * sub_blocks), 4),
                # what a per-8x4-half / per-4x4-quad walk would iterate (lower bounds: the longest sub-list of a wave, counting only
                # evaluations with a lane inside band.hi; the octagon test would put more into the lists)
                "walk_iterations_if_each_8x4_half_walked_its_own_list": sublists[0],
                "walk_iterations_if_each_4x4_quad_walked_its_own_list": sublists[1],
                "half_and_quad_sub_list_entries": [sublists[2], sublists[3]],
                "the_same_counting_only_blending_evaluations": sublists[4:8],
                "flop_model": "SURVEY 8(d): 40 flop per evaluated (pixel, splat) pair + 2*(3+7+F) per contributing pair",
                "flops": int(flops_eval), "flops_upper_bound_256R": int(flops_model),
                "TFLOP/s": round(tf, 2), "peak_TFLOP/s": VALU_PEAK_TFLOPS, "frac": round(tf / VALU_PEAK_TFLOPS, 4),
                "frac_with_256R_bound": round(flops_model / (dom_ms * launches * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4)}
        if dom == "k_render_bwd" and args.step in ("rgb", "plain") and lane_pairs:
            # K9 with geometry gradients (k_render_bwd_geo): SURVEY 8(d)'s flop model, 90 + 4 (16 + F) per contributing pair (the pairs
            # the forward blended - the backward replays exactly those) + 40 per evaluated pair for the re-evaluation of alpha
            per_c = 90.0 + 4.0 * (16 + F)
            fl = per_c * lane_pairs + 40.0 * 64 * pairs_eval
            tf = fl / (dom_ms * launches * 1e-3) / 1e12
            roof["valu"] = {"pixel_splat_pairs_contributing": lane_pairs, "pixel_splat_pairs_evaluated_by_the_forward": 64 * pairs_eval,
                            "flop_model": "SURVEY 8(d): 90 + 4*(16+F) flop per contributing pair + 40 per evaluated pair",
                            "flops": int(fl), "TFLOP/s": round(tf, 2), "peak_TFLOP/s": VALU_PEAK_TFLOPS, "frac": round(tf / VALU_PEAK_TFLOPS, 4)}
        # SURVEY 8(d)'s fraction: the larger of the HBM and the fp32 fraction of the dominant kernel
        f_hbm, f_flop = roof["hbm"]["frac"], (roof.get("valu") or {}).get("frac") or 0.0
        if f_flop > f_hbm:
            roof.update({"bound": "fp32", "achieved": roof["valu"]["TFLOP/s"], "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": f_flop})
        roof["frac_hbm"], roof["frac_fp32_flops"] = f_hbm, f_flop
        if dom == "k_render_fwd" and issue and issue.get("SQ_INSTS_VALU"):
            # secondary: how close the blend kernel is to the chip's vector-instruction ISSUE rate - the count per launch is the
            # committed PMC counter (profiles/roofline_traffic.json, taken on the tree named there), the time is this run's, the
            # peak the measured one above
            ach = issue["SQ_INSTS_VALU"] / (dom_ms * 1e-3) / 1e9
            roof["issue"] = dict({k: v for k, v in issue.items() if k != "source"}, source=issue.get("source"),
                                 achieved_G_wave_instructions_per_s=round(ach, 1), peak_plain_fp32_G_per_s=ISSUE_PEAK_PLAIN_G,
                                 issue_frac=round(ach / ISSUE_PEAK_PLAIN_G, 4),
                                 issue_frac_mix_weighted=round(ach * ISSUE_MIX_WEIGHT / ISSUE_PEAK_PLAIN_G, 4),
                                 model="achieved = SQ_INSTS_VALU per launch / this run's launch time; peak = the chip's measured rate of "
                                       "plain fp32 vector instructions (tools/micro/valu_issue.hip, profiles/r05_valu_issue.txt); "
                                       "mix_weighted prices the kernel's packed / SGPR-writing / transcendental instructions at their "
                                       "measured cost (x1.5 / x1.5 / x2.9)")
        rec["roofline"] = roof
        rec["cfg"] = cfg
    if world > 1 and args.step == "seg":
        ms = time_allreduce(cfg["P"] * cfg["F"], dev, world)
        rec["allreduce_ms"] = None if ms is None else round(ms, 3)
        P_, F_ = cfg["P"], cfg["F"]
        rec["exchange"] = {"kind": "RCCL all-reduce of the [P,F] gradient in %d row ranges" % getattr(trainer, "tail_chunks", 1),
                           "bytes": {"buffer": 4 * P_ * F_, "per_link_ring_model": int(2 * (world - 1) / world * 4 * P_ * F_)}}
        if getattr(trainer, "exchange", "rccl") in ("peer", "peer_compact") and not getattr(trainer, "sharded_tail", False):
            trainer.phase_timing = True
            trainer.step(it0 + extra_steps + 100)
            trainer.phase_timing = False
            sync()
            if trainer.last_exchange:
                rec["exchange"] = trainer.last_exchange
        elif not getattr(trainer, "sharded_tail", False):
            # per-phase device times of the multi-rank tail over a few extra (untimed) steps: how much of the collective the
            # row-range pipeline hides
            trainer.phase_timing = True
            acc = []
            for it in range(it0 + extra_steps, it0 + extra_steps + 5):
                trainer.step(it)
                if trainer.last_phases:
                    acc.append(trainer.last_phases)
            trainer.phase_timing = False
            sync()
            if acc:
                mean = lambda k: round(sum(a[k] for a in acc) / len(acc), 4)
                per = lambda k: [round(sum(a[k][i] for a in acc) / len(acc), 4) for i in range(len(acc[0][k]))]
                exposed = mean("exposed_collective_ms")
                rec["multi_rank_tail"] = {
                    "tail_chunks": acc[0]["tail_chunks"], "tail_ms": mean("tail_ms"), "gradient_kernels_ms": mean("gradient_kernels_ms"),
                    "exposed_collective_ms": exposed, "exposed_collective_ms_per_range": per("exposed_collective_ms_per_range"),
                    "optimizer_kernels_ms_per_range": per("optimizer_kernels_ms_per_range"),
                    "allreduce_alone_ms": rec["allreduce_ms"],
                    "overlap_achieved": (None if not ms else round(max(0.0, 1.0 - exposed / ms), 4)),
                    "note": "HIP events on the compute stream of rank 0, mean of 5 untimed steps; exposed = time the compute stream "
                            "stood still waiting for a row range's all-reduce"}
    del trainer
    torch.cuda.empty_cache()
    return rec


COMPACT_LIMIT = 6144      # bytes: the driver's parser lost a 24.8 KB line in round 5; the final stdout line stays below this

_CONFIG_TEXT_KEYS = ("workload", "parallelism", "arithmetic_mode", "integer_state", "multi_rank_tail")
_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_hbm", "frac_fp32_flops", "avg_launch_ms", "launches_per_step",
              "traffic", "traffic_stale", "traffic_over_algorithmic_bytes")


def compact_record(full, details_path="bench_details.json"):
    """The ONE line the driver parses: the contract keys, the numeric contract keys of `config`, the dominant kernel's roofline
    with the inputs its fractions were computed from, and the CPU baseline.  Everything else (sub_records, per-kernel tables,
    lane / issue statistics, notes) is `full`, written to ``details_path`` and to stderr by emit()."""
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    cfg = full.get("config") or {}
    c = {k: (cfg[k] if len(str(cfg[k])) <= 220 else str(cfg[k])[:217] + "...") for k in _CONFIG_TEXT_KEYS if cfg.get(k) is not None}
    for k, v in cfg.items():              # numbers and flags: views_per_s_*, roofline_frac_*, rccl_world_size, tracer, ...
        if isinstance(v, (int, float, bool)) and not isinstance(v, str):
            c[k] = v
    def lean(v):                           # a small dict of numbers: keep it, cut its prose
        if isinstance(v, dict):
            return {k: lean(x) for k, x in v.items() if k != "note"}
        return v[:120] if isinstance(v, str) else v
    for k in ("gradient_exchange", "multi_rank_tail_phases"):      # N > 1: which exchange ran, the bytes it moved, the tail's phases
        if isinstance(cfg.get(k), dict):
            c[k] = lean(cfg[k])
    out["config"] = c
    roof = full.get("roofline") or {}
    r = {k: roof.get(k) for k in _ROOF_KEYS if k in roof}
    r["bytes"] = (roof.get("hbm") or {}).get("bytes")
    wl = roof.get("workload") or {}
    r.update({k: wl.get(k) for k in ("P", "V", "R", "N", "F") if k in wl})
    valu = roof.get("valu") or {}
    if valu:
        r["pairs_evaluated"] = valu.get("pixel_splat_pairs_evaluated", valu.get("pixel_splat_pairs_evaluated_by_the_forward"))
        r["pairs_contributing"] = valu.get("pixel_splat_pairs_contributing")
        r["flops"] = valu.get("flops")
        if "lane_utilisation_of_blending_pairs" in valu:
            r["lane_utilisation_of_blending_pairs"] = valu["lane_utilisation_of_blending_pairs"]
    iss = roof.get("issue") or {}
    if iss:
        r["valu_issue_frac_of_measured_plain_fp32_rate"] = iss.get("issue_frac")
    out["roofline"] = r
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        cb = {k: (v if not isinstance(v, str) or len(v) <= 260 else v[:257] + "...") for k, v in cb.items()
              if k in ("value", "unit", "cores", "cpu", "kind", "build", "sample", "error")}
    out["cpu_baseline"] = cb
    ud = full.get("unmodified_driver")
    if isinstance(ud, dict):
        out["unmodified_driver"] = {k: v for k, v in ud.items() if isinstance(v, (int, float)) or v is None}
    out["details"] = details_path
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT:         # never again a line the driver cannot parse: shed text, keep numbers
        for k in ("sample", "build", "cpu"):
            if isinstance(out.get("cpu_baseline"), dict) and k in out["cpu_baseline"] and len(line) > COMPACT_LIMIT:
                out["cpu_baseline"][k] = str(out["cpu_baseline"][k])[:60]
                line = json.dumps(out, separators=(",", ":"))
        for k in _CONFIG_TEXT_KEYS:
            if k in out["config"] and len(line) > COMPACT_LIMIT:
                out["config"][k] = str(out["config"][k])[:60]
                line = json.dumps(out, separators=(",", ":"))
    return out


def emit(full):
    """Full record -> bench_details.json (repo root; also gpurun_out/ when that exists) and stderr; compact record -> the LAST
    line of stdout."""
    text = json.dumps(full)
    paths = [os.path.join(ROOT, "bench_details.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_details.json"))
    for p in paths:
        try:
            with open(p, "w") as f:
                f.write(text + "\n")
        except OSError as e:
            print(f"bench.py: could not write {p}: {e}", file=sys.stderr)
    print("[bench details] " + text, file=sys.stderr)
    sys.stderr.flush()
    print(json.dumps(compact_record(full), separators=(",", ":")))
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--step", default=None, choices=[None, "seg", "rgb", "plain"],
                    help="seg: train_semantic.py step (needs a feature channel: C3, C5); rgb: train.py step (C1, C2, C3); "
                         "plain (sub-records): the reference's train_semantic.py iteration on the drop-in functions alone")
    ap.add_argument("--mode", default=os.environ.get("ISR_MODE", DEFAULT_MODE), choices=["fast", "exact", "fast_reflists", "fast_tight"])
    ap.add_argument("--submodes", default="exact,fast,fast+feature_only",
                    help="at one GPU: further modes timed the same way and reported as sub_records ('' = none)")
    ap.add_argument("--more", type=int, default=1,
                    help="1 (default, C3 seg at one GPU): also time the other BASELINE configs (C2 rgb, C5 seg), the step with the "
                         "multi-view leg, a 500-step block and the plain drop-in loop, as sub_records")
    ap.add_argument("--multiview", type=int, default=0,
                    help="seg step: 1 = with the reference's multi-view leg every 10th iteration (5 more views rendered with "
                         "gradients, train_semantic.py:143-172); the headline is the single-view step, this is sub_records.C3_multiview")
    ap.add_argument("--sample-batchsize", dest="sample_batchsize", type=int, default=8192,
                    help="seg step: samples per loss (BASELINE config 3 names 8 192; the reference's default is 32 768)")
    ap.add_argument("--feat-dim", dest="feat_dim", type=int, default=None, help="seg step: feature width instead of the config's")
    ap.add_argument("--empty-cache", dest="empty_cache", type=int, default=0, help="plain step: torch.cuda.empty_cache() every iteration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tracer", type=int, default=1, help="produce gau_related_pixels each forward like the reference")
    ap.add_argument("--async-binning", type=int, default=1,
                    help="size a view's binning workspace from its own verified count of an earlier forward instead of a "
                         "blocking read of R")
    ap.add_argument("--view-cache-gb", type=float, default=0.0,
                    help="opt-in exploration, NOT the headline configuration: keep each view's geometry pass + binning "
                         "while the geometry is frozen (rasterizer.set_view_cache); 0 = recompute every step like the "
                         "reference")
    ap.add_argument("--lazy-maps", type=int, default=0,
                    help="1: evaluate render()'s seven derived normal/depth maps on first access (the seg step never reads "
                         "them) instead of inside render() like the reference (default 0 = reference behaviour)")
    ap.add_argument("--spatial-sort", type=int, default=1,
                    help="1 (default): the trainer stores the Gaussians in Z-order of their centres (sorted once at load, "
                         "before the timed region; a pure relabelling of rows); 0: keep the generator's random order")
    ap.add_argument("--fused-sampling", type=int, default=1,
                    help="1 (default): one kernel draws every index of a step (iso_sample_step); 0: torch.randint + gathers")
    ap.add_argument("--exchange", default=os.environ.get("ISR_EXCHANGE", "rccl"), choices=["rccl", "peer", "peer_compact"],
                    help="N > 1, seg step: how dL/dparam is summed - RCCL all-reduce (default), or the direct exchange over peer-mapped "
                         "buffers, dense or compacted to the touched rows (peer_exchange.PeerExchange; opt-in, untimed on a multi-GPU node)")
    ap.add_argument("--sharded-tail", type=int, default=None,
                    help="1: several ranks exchange dL/dparam by reduce-scatter, run Adam on their shard of the rows and all-gather "
                         "the parameter rows (SegTrainer.sharded_tail; default: ISR_SHARDED_TAIL or off)")
    ap.add_argument("--split-tail", type=int, default=0,
                    help="1: with one rank, take the multi-rank form of the per-Gaussian tail to measure what it costs")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(args.gpus)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # ISR_DIST_BACKEND=gloo (testing only): several ranks may then share one GPU, which RCCL does not allow
    backend = os.environ.get("ISR_DIST_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        print(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s); RCCL needs one GPU per rank", file=sys.stderr)
        sys.exit(2)
    local = local if backend == "nccl" else local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from instascene_amd import scenes
    cfg0 = scenes.CONFIGS[args.config]
    if args.step is None:
        args.step = "seg" if cfg0["F"] > 0 else "rgb"
    if args.step == "seg" and cfg0["F"] == 0:
        print(f"bench.py: --step seg needs a feature channel; {args.config} has none (use --step rgb)", file=sys.stderr)
        sys.exit(2)

    head = run(args, args.mode, rank, world, dev, detail=True)
    subs = {}

    def sub(name, mode=None, repeats=1, note=None, **over):
        """One more measurement under the same clock, reported as sub_records[name]: same timing rules (warm-up, then a
        block of steps between synchronisations), its own dominant kernel / roofline fraction / instance count."""
        a = argparse.Namespace(**vars(args))
        for k, v in over.items():
            setattr(a, k, v)
        # every record starts from a compact allocator: what the previous record left cached or prefetched (C5-size states,
        # two views ahead) is dropped - the plain loop's empty_cache() variant otherwise pays for returning it
        import gc
        import torch
        from instascene_amd import rasterizer as _rz
        torch.cuda.synchronize()
        _rz._PREFETCHED.clear()
        gc.collect()
        torch.cuda.empty_cache()
        try:
            r = run(a, mode or a.mode, rank, world, dev, detail=True, repeats=repeats)
        except Exception as e:       # the headline must still be printed
            subs[name] = {"error": repr(e)}
            import torch
            torch.cuda.empty_cache()
            return
        roof = r.pop("roofline", None) or {}
        cfg_ = r.pop("cfg", None) or {}
        r["workload"] = "%s %s step: P=%s, %sx%s, F=%s" % (a.config, a.step, cfg_.get("P"), cfg_.get("W"), cfg_.get("H"),
                                                           roof.get("workload", {}).get("F"))
        r["dominant_kernel"] = roof.get("kernel")
        r["dominant_kernel_ms"] = roof.get("avg_launch_ms")
        r["dominant_kernel_launches_per_step"] = roof.get("launches_per_step")
        r["dominant_kernel_roofline_frac"] = roof.get("frac")        # max(HBM, fp32) fraction
        r["dominant_kernel_frac_hbm"] = roof.get("frac_hbm")
        r["dominant_kernel_frac_fp32_flops"] = roof.get("frac_fp32_flops")
        if "valu" in roof:
            r["dominant_kernel_flop_model"] = {k: roof["valu"].get(k) for k in ("flop_model", "flops", "TFLOP/s", "pixel_splat_pairs_contributing")}
        r["R"] = roof.get("workload", {}).get("R")
        r["V"] = roof.get("workload", {}).get("V")
        r["kernels_ms_x_launches_per_step"] = {k: [v["ms_per_launch"], v["launches_per_view"]] for k, v in roof.get("kernels", {}).items()}
        if repeats > 1:
            r["timing"] = "the faster of %d consecutive blocks of %d steps" % (repeats, a.steps)
        if note:
            r["note"] = note
        subs[name] = r

    if world == 1:
        parity = {"exact": "radii / tiles_touched / point_list / ranges / n_contrib / images bit-identical to the CPU oracle",
                  "fast_reflists": "FAST arithmetic on the reference's rectangles: radii / tiles_touched / point_list / ranges "
                                   "bit-identical to the reference's; images 1e-4",
                  "fast_tight": "= fast",
                  "fast": "tile lists are order-preserving subsequences of the reference's; outputs = fast_reflists' bits except "
                          "the distortion channel's last ones",
                  "fast+feature_only": "opt-in pipe.feature_only_forward (NOT the reference's behaviour: render() returns no colour / "
                                       "depth / normal maps and no tracer list); feature map, binning and the step's parameters "
                                       "bit-identical to `fast`"}
        for m in [m for m in args.submodes.split(",") if m and m != args.mode]:
            if m.endswith("+feature_only") and args.step != "seg":
                continue
            sub(m, mode=m, repeats=2, note=parity.get(m))
        if args.more and args.config == "C3" and args.step == "seg":
            # the other BASELINE configurations and the other loops, in the same process under the same clock
            # (the plain loop first: its empty_cache() variant returns and re-requests device memory every iteration, and how
            # long the driver takes for that depends on what the process has allocated before - 17 ms per iteration in a fresh
            # process, up to 90 ms after the C5-size records)
            plain_note = ("harness.PlainSegTrainer: the reference's iteration as the reference writes it (train_semantic.py:95-208) on "
                          "render() and contrastive_loss() alone, library defaults (blocking instance-count read, tracer on), "
                          "multi-view leg every 10th iteration, torch.optim.Adam; 30 steps = 3 multi-view iterations")
            sub("dropin_plain_fast", mode=DEFAULT_MODE, step="plain", steps=30, warmup=10, note=plain_note + "; the drop-in's default mode (fast_reflists)")
            sub("dropin_plain_exact", mode="exact", step="plain", steps=30, warmup=10, note=plain_note + "; ISR_MODE=exact")
            from instascene_amd import dropin as _dropin
            _dropin.empty_cache_under_pressure()          # what dropin.install() does for the unmodified driver
            sub("dropin_plain_fast_empty_cache", mode=DEFAULT_MODE, step="plain", steps=30, warmup=10, empty_cache=True,
                note=plain_note + "; plus torch.cuda.empty_cache() every iteration like the reference (:206), under the drop-in as "
                     "installed: the library's buffers live in its own arena (arena.py) and install() makes empty_cache() act only "
                     "under memory pressure (dropin.empty_cache_under_pressure)")
            _dropin.restore_empty_cache()
            sub("dropin_plain_fast_empty_cache_honoured", mode=DEFAULT_MODE, step="plain", steps=30, warmup=10, empty_cache=True,
                note=plain_note + "; ISR_KEEP_EMPTY_CACHE=1: torch's own empty_cache() every iteration - the library's buffers stay in "
                     "the arena, the reference's torch temporaries (~9 device allocations per iteration) go back to the driver and "
                     "are hipMalloc-ed again; this record measures the driver (it slows down with the process's allocation count)")
            sub("soak_500", steps=500, warmup=5, note="the headline configuration, one block of 500 steps")
            sub("C3_multiview", steps=40, warmup=10, multiview=True,
                note="the reference's default step: the multi-view leg (5 more views rendered with gradients through the dense "
                     "[F,H,W] feature map, train_semantic.py:143-172, lambda_multiview_contras = 1e-6) every 10th iteration; "
                     "40 steps = 4 such iterations, ms_per_step is their mean")
            sub("C2_rgb", config="C2", step="rgb", steps=200, warmup=10, note="BASELINE config 2: the train.py step")
            sub("C3_rgb", config="C3", step="rgb", steps=50, warmup=5, note="the train.py step at C3 size")
            sub("C5_seg", config="C5", step="seg", steps=50, warmup=5, note="BASELINE config 5 in its 1-GPU form (F = 64: one 64-channel feature pass)")
            sub("reference_defaults", steps=40, warmup=10, multiview=True, sample_batchsize=32 * 1024, feat_dim=16,
                note="the reference's default training configuration at C3 size (arguments/__init__.py:65,103-104): seg_feat_dim = 16, "
                     "sample_batchsize = 32 768, multi-view leg on (every 10th iteration); 40 steps = 4 such iterations")

    if rank == 0:
        cfg = head.pop("cfg")
        roof = head.pop("roofline")
        if args.step == "seg":
            metric = "train-step views/sec (fwd+bwd) @1.5M Gaussians, 1080p, 32-d feat"
            if args.config != "C3":
                metric = f"train-step views/sec (fwd+bwd), {args.config}"
            workload = (f"{args.config}: {cfg['P']} Gaussians, {cfg['W']}x{cfg['H']}, F={cfg['F']}, sample batch 8192, "
                        "2 single-view + 1 3-D contrastive loss, Adam on [P,F] (train_semantic.py step)")
        else:
            metric = f"train-step views/sec (fwd+bwd), train.py step, {args.config}"
            workload = (f"{args.config}: {cfg['P']} Gaussians, {cfg['W']}x{cfg['H']}, RGB + depth + normal, L1 + SSIM + normal "
                        "consistency, full geometry backward, Adam on six parameter groups (train.py step)")
        par = f"dp{world} (one view per rank"
        if world > 1:
            par += (", RCCL all-reduce of the [P,F] gradient in row ranges overlapped with the per-Gaussian tail and the next view's "
                    "geometry pass)" if args.step == "seg" else ", all-reduce of the six parameter groups' gradients)")
        else:
            par += ")"
        out = {"metric": metric, "value": head["value"], "unit": "views/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": workload, "parallelism": par, "arithmetic_mode": args.mode,
                          "parity_of_this_mode": {"fast": "per-pixel decisions are EXACT's by construction; images within 1e-4, gradients "
                                                          "within 1e-3 of the oracle, gated by cause (tests/test_gpu_rasterizer.py, "
                                                          "tests/test_gpu_fuzz.py); tile lists are order-preserving subsequences of the "
                                                          "reference's (a splat is binned where its alpha >= 1/255 box reaches) - every "
                                                          "output equals fast_reflists' bit for bit except the distortion channel's last bits",
                                                  "exact": "every forward output and all integer state bit-identical to the CPU oracle",
                                                  "fast_reflists": "as fast, on the reference's rectangles: radii, tiles_touched, point_list, "
                                                                   "ranges bit-identical to the reference's",
                                                  "fast_tight": "= fast"}[args.mode],
                          "tracer": bool(args.tracer), "async_binning": bool(args.async_binning),
                          "view_cache_gb": args.view_cache_gb,
                          "rccl_world_size": (dist.get_world_size() if world > 1 else 1),
                          "allreduce_ms_per_step_alone": head.get("allreduce_ms"),
                          "multi_rank_tail_phases": head.get("multi_rank_tail"),
                          "gradient_exchange": head.get("exchange"),
                          "gaussian_order": "z-order of the centres, sorted once at load" if args.spatial_sort else "as generated (random)",
                          "derived_render_maps": "on first access (never read by the seg step)" if args.lazy_maps else "inside render(), like the reference",
                          "multi_rank_tail": ("sharded: reduce-scatter, owner-only Adam, all-gather of the parameter rows"
                                              if head.get("sharded_tail") else
                                              "replicated: row-range pipelined all-reduce, Adam on every rank") if world > 1 else "n/a (one rank)",
                          "step_loop_stream": "the trainer's own high-priority HIP stream for the whole block of steps (SegTrainer.stream_scope); "
                                              "the next view's geometry pass + binning on one side stream",
                          "view_order": "a seeded random permutation of the 16 ring cameras per epoch (dist_utils.view_order) = the "
                                        "reference's random pop from a refilled stack (train_semantic.py:96-100) with the draws made up "
                                        "front: the coming views are known ahead, so the geometry pass + binning of the view TWO steps "
                                        "ahead are issued on a side stream during the current step (SegTrainer.prefetch_distance; every "
                                        "step still runs one full chain - the main stream just never waits for it)",
                          "hoisted_out_of_the_timed_region": (
                              ["activations of the frozen parameters (exp / sigmoid / normalize, SH concat): evaluated once",
                               "per-view pools of labelled pixels and of visible labelled Gaussians, the cameras' ray tables, each "
                               "view's verified tile-instance count (SegTrainer.warm_view_caches: one untrained render per view)",
                               "one priming pass over the 16 views whose effect on parameters / optimiser / RNG is undone (SegTrainer.prime: code objects, "
                               "allocator and arena pools at their steady-state size, GPU clocks up - a fresh process otherwise spends its first ~20 "
                               "steps 3-5 % slow)",
                               "render()'s `visibility_filter` (radii > 0, one elementwise kernel) and the tracer list's slice are "
                               "evaluated on first access of the dict entry; a warmed-up step reads neither",
                               "the next step's index draw (a function of seed, iteration and view) is issued behind this step's forward"]
                              if args.step == "seg" else
                              ["targets = initial renders + noise", "the cameras' ray tables, each view's verified tile-instance count"])},
               "roofline": roof}
        # the contract numbers as NUMERIC keys of `config` (the driver's record keeps `config`, not `sub_records`)
        val = lambda k: (subs.get(k) or {}).get("value")
        out["config"].update({
            "integer_state": {"fast": "tiles_touched / point_list / ranges / num_rendered / n_contrib are order-preserving "
                                      "subsequences of the reference's (same output bits as on the reference's lists)",
                              "fast_tight": "as fast", "fast_reflists": "the reference's, bit for bit",
                              "exact": "the reference's, bit for bit"}[args.mode],
            "views_per_s_exact": val("exact"),
            "views_per_s_reference_tile_lists": head["value"] if args.mode == "fast_reflists" else val("fast_reflists"),
            "views_per_s_tight_tile_lists": head["value"] if args.mode in ("fast", "fast_tight") else val("fast"),
            "views_per_s_unmodified_driver": val("dropin_plain_fast"),
            "views_per_s_unmodified_driver_with_empty_cache": val("dropin_plain_fast_empty_cache"),
            "views_per_s_C2_rgb": val("C2_rgb"), "views_per_s_C3_rgb": val("C3_rgb"), "views_per_s_C5_seg": val("C5_seg"),
            "views_per_s_C3_multiview": val("C3_multiview"), "views_per_s_reference_defaults": val("reference_defaults"),
            "views_per_s_soak_500": val("soak_500"),
            "roofline_frac_C2_rgb_k_render_bwd_geo": (subs.get("C2_rgb") or {}).get("dominant_kernel_roofline_frac"),
            "roofline_frac_C5_seg": (subs.get("C5_seg") or {}).get("dominant_kernel_roofline_frac")})
        if subs.get("dropin_plain_fast") or subs.get("dropin_plain_fast_empty_cache"):
            # what a maintainer gets who installs the drop-in and runs train_semantic.py unmodified (harness.PlainSegTrainer;
            # details in sub_records): next to the headline, not only inside the long sub_records object
            ud = {"views_per_s": (subs.get("dropin_plain_fast") or {}).get("value"),
                  "views_per_s_with_the_reference's_empty_cache_every_iteration": (subs.get("dropin_plain_fast_empty_cache") or {}).get("value"),
                  "workspace": "library-owned arena (instascene_amd/arena.py): empty_cache() frees none of the per-forward buffers"}
            out = dict(list(out.items())[:9] + [("unmodified_driver", ud)] + list(out.items())[9:])
        if subs:
            out["sub_records"] = subs
        if world > 1:
            out["cpu_baseline"] = None       # timed on rank 0 at N=1 only (task contract)
        elif not args.no_cpu_baseline:
            try:
                cfg["index"] = scenes.CONFIGS[args.config]["index"]
                out["cpu_baseline"] = cpu_baseline(cfg, args.step, 16384)
            except Exception as e:   # the bench line must still be printed
                out["cpu_baseline"] = {"error": repr(e)}
        emit(out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
