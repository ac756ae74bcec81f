#!/usr/bin/env python
"""Where does the host spend a C3 train step?  cProfile over K steps of the headline trainer (tottime / cumtime tables).
usage: host_profile.py [--steps K] [--sort tottime|cumtime] [--top N]"""
import argparse, cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instascene_amd import scenes, rasterizer
from instascene_amd.harness import SegTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--top", type=int, default=45)
ap.add_argument("--scale", type=float, default=1.0, help="shrink the scene: a small one makes the step host-bound")
ap.add_argument("--profile", type=int, default=1)
ap.add_argument("--st-autograd", type=int, default=0, help="run the backward on the calling thread")
a = ap.parse_args()
rasterizer.set_mode(os.environ.get("ISR_MODE", "fast_reflists")); rasterizer.set_tracer(True); rasterizer.set_async_binning(True)
scene, cams, cfg = scenes.config_scene("C3", a.scale)
tr = SegTrainer(scene, cams[:16], device="cuda", sample_batchsize=8192, use_class_feat=True)
tr.warm_view_caches()
if a.st_autograd:
    torch.autograd.set_multithreading_enabled(False)
tr.prime(steps=16)
with tr.stream_scope():
    for it in range(32):
        tr.step(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(32, 32 + a.steps):
        tr.step(it)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"unprofiled: host enqueue {(t1 - t0) / a.steps * 1e3:.3f} ms/step, total {(t2 - t0) / a.steps * 1e3:.3f} ms/step")
    if not a.profile:
        sys.exit(0)
    pr = cProfile.Profile()
    base = 32 + a.steps
    pr.enable()
    for it in range(base, base + a.steps):
        tr.step(it)
    pr.disable()
    torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(a.top)
    print(f"==== by {key} (over {a.steps} steps) ====")
    print(s.getvalue())
