#!/usr/bin/env python
"""Long-run check of a trainer: step time, loss and workload every BLOCK iterations - what found item 34 of docs/history/DESIGN_rounds_1-3.md section 3
(a train.py-style run slowing 2.4x over 3 000 iterations because sixteen surfels had grown over the whole view).
usage: soak_train.py [--config C2|C3] [--step rgb|seg|plain] [--blocks 6] [--block 500] [--scale 1.0] [--empty-cache 1]
(--step plain: harness.PlainSegTrainer = the reference's unmodified train_semantic.py iteration on the drop-in functions; --empty-cache 1:
with its torch.cuda.empty_cache() every iteration (torch's own), 2: the same under dropin.install()'s empty_cache policy)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instascene_amd import scenes, rasterizer
from instascene_amd.harness import RgbTrainer, SegTrainer, PlainSegTrainer
from instascene_amd import arena
from instascene_amd.render import render

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2")
ap.add_argument("--step", default=None)
ap.add_argument("--blocks", type=int, default=6)
ap.add_argument("--block", type=int, default=500)
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--empty-cache", dest="empty_cache", type=int, default=0)
ap.add_argument("--densify", type=int, default=0, help="rgb step: 1 = density control on (clone / split / prune every 100 iterations)")
a = ap.parse_args()
rasterizer.set_mode(os.environ.get("ISR_MODE", "fast_reflists")); rasterizer.set_tracer(True); rasterizer.set_async_binning(a.step != "plain")
scene, cams, cfg = scenes.config_scene(a.config, a.scale)
step = a.step or ("seg" if cfg["F"] > 0 else "rgb")
dev = torch.device("cuda")
if step == "rgb":
    scene.seg_feature = None
    H, W = cams[0].image_height, cams[0].image_width
    dens = dict(from_iter=100, until_iter=10 ** 6, interval=100, opacity_reset_interval=3000, grad_threshold=0.00002) if a.densify else None
    tr = RgbTrainer(scene, cams[:16], [torch.zeros(3, H, W)] * 16, device="cuda", densify=dens)
    g = torch.Generator(device=dev).manual_seed(5)
    with torch.no_grad():
        tr.targets = [(render(c, tr.model, tr.pipe, tr.bg)["render"].detach() + 0.05 * torch.randn(3, H, W, device=dev, generator=g)).clamp(0, 1)
                      for c in tr.cams]
    run = lambda it: tr.step(it)[0]
elif step == "plain":
    if a.empty_cache == 2:          # under the drop-in as installed
        from instascene_amd import dropin
        dropin.empty_cache_under_pressure()
    tr = PlainSegTrainer(scene, cams[:16], device="cuda", sample_batchsize=8192, empty_cache=bool(a.empty_cache))
    run = lambda it: tr.step(it)
else:
    tr = SegTrainer(scene, cams[:16], device="cuda", sample_batchsize=8192, use_class_feat=True)
    tr.warm_view_caches()
    tr.prime()
    run = lambda it: tr.step(it)
for it in range(10):
    run(it)
torch.cuda.synchronize()
it = 10
for blk in range(a.blocks):
    t0 = time.perf_counter()
    for _ in range(a.block):
        loss = run(it); it += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.block * 1e3
    print("iterations %6d: %.3f ms/step  loss %.4f  P %d  tile instances (last view) %d  reserved %.2f GB  allocator growth %d  arena %.2f GB in %d blocks (%d leases)" % (
        it, dt, float(loss.detach()), tr.model._xyz.shape[0], rasterizer.LAST_NUM_RENDERED, torch.cuda.memory_reserved() / 2 ** 30,
        torch.cuda.memory_stats().get("num_device_alloc", 0), arena.reserved_bytes() / 2 ** 30, arena.STATS["new_blocks"] - arena.STATS["dropped_blocks"],
        arena.STATS["leases"]), flush=True)
if step == "rgb":
    with torch.no_grad():
        r = render(tr.cams[0], tr.model, tr.pipe, tr.bg)["radii"].float()
    print("screen radii of view 0: max %.0f px, %d above 128 px, %d above 256 px" % (float(r.max()), int((r > 128).sum()), int((r > 256).sum())))
