#!/usr/bin/env python
"""train.py-style step: render + L1/SSIM/normal losses + full geometry backward + Adam on all six
parameter groups.  Secondary benchmark (the headline metric is bench.py).  usage: bench_rgb.py [--config C3] [--steps K]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instascene_amd import scenes, rasterizer
from instascene_amd._lib import lib
from instascene_amd.harness import RgbTrainer
from instascene_amd.render import render

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--mode", default="fast")
a = ap.parse_args()
rasterizer.set_mode(a.mode); rasterizer.set_tracer(False); rasterizer.set_async_binning(True)
scene, cams, cfg = scenes.config_scene(a.config)
scene.seg_feature = None
cams = cams[:8]
tr = RgbTrainer(scene, cams, [torch.zeros(3, cfg["H"], cfg["W"])] * len(cams), device="cuda")
with torch.no_grad():      # targets: the initial renders plus noise
    tr.targets = [(render(c, tr.model, tr.pipe, tr.bg)["render"] + 0.05 * torch.randn(3, cfg["H"], cfg["W"], device="cuda")).clamp(0, 1)
                  for c in tr.cams]
L = lib()
for it in range(a.warmup):
    tr.step(it)
torch.cuda.synchronize()
L.isr_profile_enable(1)
t0 = time.perf_counter()
for it in range(a.warmup, a.warmup + a.steps):
    tr.step(it)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
import ctypes
buf = ctypes.create_string_buffer(1 << 14); L.isr_profile_summary(buf, len(buf)); L.isr_profile_enable(0)
kern = {}
for ln in buf.value.decode().splitlines():
    n, c, t = ln.split()
    kern[n] = round(float(t) / int(c), 4)
print(json.dumps({"metric": "train.py-style step (RGB + geometry, full backward)", "config": a.config, "mode": a.mode,
                  "P": cfg["P"], "W": cfg["W"], "H": cfg["H"], "ms_per_step": round(1e3 * dt / a.steps, 3),
                  "views_per_s": round(a.steps / dt, 2), "kernels_ms_per_launch": kern}))
