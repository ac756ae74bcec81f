#!/usr/bin/env python
"""Benchmark of the hot path named by BASELINE.json: train-step views/s (render forward + backward,
three contrastive losses, Adam on the [P,F] feature) on the synthetic C3 workload
(1.5 M Gaussians, 1920x1080, 32-d feature, sample batch 8192) — SURVEY.md §8(d).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W

One rank per GPU; every rank renders a different view per step (weak scaling), the [P,F] gradient is
summed with an RCCL all-reduce.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def algorithmic_bytes(P, V, R, N, F):
    """algorithmic HBM bytes per view, per kernel group."""
    fwd_blend = R * (64 + 12 + 4 * F) + N * (40 + 20 + 4 * F)          # K8: staged records/colours/features + per-pixel outputs
    bwd_blend = N * (40 + 20 + 4 * F) + R * (64 + 12 + 4 * F) + R * 2 * 4 * F   # K9 (feature-only): grads+state, staging, one row write+read
    pre = 12 * P + 307 * V + 8 * P
    binning = R * (12 + 8 + 12)                                          # bucket write, sort read, list write (this design; reference: 164 B)
    return dict(k_render_fwd=fwd_blend, k_render_bwd=bwd_blend, k_preprocess=pre, binning=binning)


def profile_summary(L):
    buf = ctypes.create_string_buffer(1 << 16)
    n = L.isr_profile_summary(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, tot = line.split()
        out[name] = (int(cnt), float(tot))
    return out


def cpu_baseline(cfg, seconds_budget=25.0):
    """The CPU oracle (C++/OpenMP restatement of the reference kernels) timed on this box's host cores on a
    bounded sample of the same workload: forward + full backward of one C3 view, at the largest scale
    (1, 1/2 or 1/4 in each image dimension, P scaled alike) whose estimated time fits the budget."""
    import numpy as np
    import oracle
    from instascene_amd import scenes

    def build(scale):
        P = int(cfg["P"] * scale * scale)
        W, H = int(cfg["W"] * scale), int(cfg["H"] * scale)
        sc = scenes.synthetic_scene(P, cfg["F"], scenes.SEED_BASE + 3, cfg["mu_s"] - math.log(scale))
        cam = scenes.ring_cameras(64, W, H)[0]
        a = {k: (None if v is None else v.numpy()) for k, v in scenes.activated_inputs(sc).items()}
        return P, W, H, cam, a

    def one(P, W, H, cam, a):
        st = oracle.forward(a["means3D"], a["opacities"], cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(),
                            cam.camera_center.numpy(), np.zeros(3, np.float32), W, H, math.tan(cam.FoVx / 2),
                            math.tan(cam.FoVy / 2), scales=a["scales"], rotations=a["rotations"], shs=a["shs"],
                            extra=a["extra"], sh_degree=3)
        oracle.backward(st, np.zeros_like(st["color"]), np.zeros_like(st["others"]), np.ones_like(st["extra"]))
        return st["R"]

    # calibrate on the 1/4-scale view (splats enlarged by 1/scale in world units: same pixel footprint and depth complexity)
    small = build(0.25)
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
    return {"value": frac / dt, "unit": "views/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": f"oracle forward + full backward of a C3 view at scale {scale:g} (P={P}, {W}x{H}, F={cfg['F']}, R={R}): "
                      f"{n} view(s) in {dt * n:.1f} s" + ("" if scale == 1.0 else
                      f"; value = measured {1.0 / dt:.3f} views/s x {frac:g} (work scales with P and pixels)")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--mode", default=os.environ.get("ISR_MODE", "fast"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tracer", type=int, default=1, help="produce gau_related_pixels each forward like the reference")
    ap.add_argument("--async-binning", type=int, default=1,
                    help="size the binning workspace from the previous view instead of a blocking read of R")
    ap.add_argument("--view-cache-gb", type=float, default=0.0,
                    help="opt-in exploration, NOT the headline configuration: keep each view's geometry pass + binning "
                         "while the geometry is frozen (rasterizer.set_view_cache); 0 = recompute every step like the "
                         "reference")
    ap.add_argument("--lazy-maps", type=int, default=0,
                    help="1: evaluate render()'s seven derived normal/depth maps on first access (this step never reads "
                         "them) instead of inside render() like the reference (default 0 = reference behaviour)")
    ap.add_argument("--spatial-sort", type=int, default=1,
                    help="1 (default): the trainer stores the Gaussians in Z-order of their centres (sorted once at load, "
                         "before the timed region; a pure relabelling of rows); 0: keep the generator's random order")
    ap.add_argument("--fused-sampling", type=int, default=1,
                    help="1 (default): one kernel draws every index of a step (iso_sample_step); 0: torch.randint + gathers")
    ap.add_argument("--split-tail", type=int, default=0,
                    help="1: with one rank, take the multi-rank form of the per-Gaussian tail (dL/dx kernel, [no-op] "
                         "all-reduce, Adam kernel) to measure what it costs next to the one-pass tail")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("launch with torch.distributed.run for --gpus > 1", file=sys.stderr)
        sys.exit(2)
    # ISR_DIST_BACKEND=gloo (testing only): several ranks may then share one GPU, which RCCL does not allow
    backend = os.environ.get("ISR_DIST_BACKEND", "nccl")
    local = local if backend == "nccl" else local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from instascene_amd import scenes, rasterizer
    from instascene_amd._lib import lib
    from instascene_amd.harness import SegTrainer

    rasterizer.set_mode(args.mode)
    rasterizer.set_tracer(bool(args.tracer))
    rasterizer.set_async_binning(bool(args.async_binning))
    rasterizer.set_view_cache(args.view_cache_gb)
    scene, cams, cfg = scenes.config_scene(args.config)
    if cfg["F"] == 0:
        print("bench.py measures the feature-training step: the config needs F > 0 (C3, C5); the RGB + geometry step of "
              "C1 / C2 is tools/bench_rgb.py", file=sys.stderr)
        sys.exit(2)
    trainer = SegTrainer(scene, cams[:16], device=dev, sample_batchsize=8192, use_class_feat=True, rank=rank, world=world,
                         spatial_sort=bool(args.spatial_sort), fused_sampling=bool(args.fused_sampling))
    trainer.split_tail = bool(args.split_tail)
    trainer.pipe.lazy_maps = bool(args.lazy_maps)
    trainer.warm_view_caches()       # per-view constants (ray tables, visible pools): setup, like the label maps
    trainer.prime()                  # code objects, allocator pools, side stream: two steps whose effect is undone
    L = lib()

    for it in range(args.warmup):
        trainer.step(it)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    L.isr_profile_enable(2)          # HIP events around the dominant kernel only inside the timed region
    t0 = time.perf_counter()
    for it in range(args.warmup, args.warmup + args.steps):
        trainer.step(it)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof_dom = profile_summary(L)
    # every kernel of the library, over a few extra (untimed) steps: detail for the JSON line
    L.isr_profile_enable(1)
    extra_steps = min(5, args.steps)
    for it in range(args.warmup + args.steps, args.warmup + args.steps + extra_steps):
        trainer.step(it)
    torch.cuda.synchronize()
    prof_all = profile_summary(L)
    L.isr_profile_enable(0)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        # workload statistics of the last rendered view, for the byte model
        from instascene_amd.render import render
        with torch.no_grad():
            vi = trainer.view_index(args.warmup + args.steps - 1)
            pkg = render(trainer.cams[vi], trainer.model, trainer.pipe, trainer.bg)
            V = int((pkg["radii"] > 0).sum().item())
        P, N, F = cfg["P"], cfg["W"] * cfg["H"], cfg["F"]
        ptr = ctypes.c_int64(0)
        # R of that view: re-run prepare is not needed — total instances = sum of tiles touched == binning size
        R = int(rasterizer.LAST_NUM_RENDERED) if hasattr(rasterizer, "LAST_NUM_RENDERED") else 0
        ab = algorithmic_bytes(P, V, R, N, F)
        kern_ms = {k: (tot / cnt) for k, (cnt, tot) in prof_all.items()}
        dom = max(kern_ms, key=lambda k: kern_ms[k] * prof_all[k][0]) if kern_ms else None
        if dom in prof_dom:             # measured over the timed region itself
            prof, steps_prof = prof_dom, args.steps
            kern_ms[dom] = prof_dom[dom][1] / prof_dom[dom][0]
        else:
            prof, steps_prof = prof_all, extra_steps
        roof = None
        if dom is not None:
            per_launch_bytes = ab.get(dom, 0)
            launches_per_view = prof[dom][0] / float(steps_prof)
            achieved = per_launch_bytes / max(launches_per_view, 1e-9) / (kern_ms[dom] * 1e-3) / 1e9 if per_launch_bytes else 0.0
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(dom)
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                    "avg_launch_ms": round(kern_ms[dom], 4), "launches_per_view": round(launches_per_view, 3),
                    "algorithmic_bytes_per_view": int(per_launch_bytes),
                    "note": "blend kernels are VALU/LDS-bound, not HBM-bound (SURVEY 8d); per-kernel ms below",
                    "kernels_ms_per_launch": {k: round(v, 4) for k, v in sorted(kern_ms.items())},
                    "kernels_launches_per_view": {k: round(prof_all[k][0] / float(extra_steps), 2) for k in sorted(prof_all)},
                    "timing": "HIP events on the launch stream: dominant kernel over the timed region, the others "
                              "over %d extra untimed steps" % extra_steps,
                    "workload": {"P": P, "V": V, "R": R, "N": N, "F": F}}
        out = {"metric": "train-step views/sec (fwd+bwd) @1.5M Gaussians, 1080p, 32-d feat",
               "value": round(world * args.steps / dt, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{args.config}: {cfg['P']} Gaussians, {cfg['W']}x{cfg['H']}, F={cfg['F']}, "
                                      f"sample batch 8192, 2 single-view + 1 3-D contrastive loss, Adam on [P,F]",
                          "parallelism": f"dp{world} (one view per rank, RCCL all-reduce of the [P,F] gradient"
                                         + (", overlapped with the next view's geometry pass)" if world > 1 else ")"),
                          "arithmetic_mode": args.mode, "tracer": bool(args.tracer),
                          "async_binning": bool(args.async_binning), "view_cache_gb": args.view_cache_gb,
                          "gaussian_order": "z-order of the centres, sorted once at load" if args.spatial_sort else "as generated (random)",
                          "derived_render_maps": "on first access (never read by this step)" if args.lazy_maps else "inside render(), like the reference"},
               "roofline": roof}
        if world > 1:
            out["cpu_baseline"] = None       # timed on rank 0 at N=1 only (task contract)
        elif not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg)
            except Exception as e:   # the bench line must still be printed
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
