#!/usr/bin/env python
"""How long does the host need to ENQUEUE a train step vs how long the GPU needs to run it?
usage: host_overhead.py [--lazy-maps 0|1] [--steps K]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instascene_amd import scenes, rasterizer
from instascene_amd.harness import SegTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--lazy-maps", type=int, default=0)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--delay-us", type=float, default=0.0, help="busy-wait on the host after render() returns")
ap.add_argument("--delay-at", default="render")
ap.add_argument("--scale", type=float, default=1.0, help="shrink the scene: the step becomes host-bound")
a = ap.parse_args()
rasterizer.set_mode(os.environ.get("ISR_MODE", "fast_reflists")); rasterizer.set_tracer(True); rasterizer.set_async_binning(True)
scene, cams, cfg = scenes.config_scene("C3", a.scale)
tr = SegTrainer(scene, cams[:16], device="cuda", sample_batchsize=8192, use_class_feat=True)
tr.pipe.lazy_maps = bool(a.lazy_maps)
tr.warm_view_caches()
from instascene_amd import harness as _h
_ev = []
_host_t = []
def _mark(tag):
    e = torch.cuda.Event(enable_timing=True); e.record(); _ev.append((tag, e)); _host_t.append(time.perf_counter())
_orig_render, _orig_ar = _h.render, _h.allreduce_grads
def _spin(us):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e6 < us:
        pass
def _render(*x, **k):
    if a.delay_at == "before": _spin(a.delay_us)
    _mark("start"); r = _orig_render(*x, **k)
    if a.delay_at == "render": _spin(a.delay_us)
    _mark("render: maps"); return r
def _ar(*x, **k):
    _mark("losses+backward"); return _orig_ar(*x, **k)
_h.render, _h.allreduce_grads = _render, _ar
_orig_cl = _h.contrastive_loss
_ncl = [0]
def _cl(*x, **k):
    r = _orig_cl(*x, **k)
    _ncl[0] += 1
    if _ncl[0] % 3 == 0: _mark("losses fwd")
    return r
_h.contrastive_loss = _cl
_orig_rb = rasterizer.rasterize_gaussians_backward
def _rb(*x, **k):
    _mark("loss bwd (contrastive, gather)")
    r = _orig_rb(*x, **k)
    _mark("raster bwd")
    return r
rasterizer.rasterize_gaussians_backward = _rb
_orig_prep, _orig_rg = rasterizer._prepare, rasterizer.rasterize_gaussians
def _prep(*x, **k):
    _mark("render: before geometry pass (normalise, allocs)")
    r = _orig_prep(*x, **k)
    _mark("render: geometry pass")
    return r
def _rg(*x, **k):
    r = _orig_rg(*x, **k)
    _mark("render: binning + blend")
    _mark("(two marks back to back)")
    return r
from instascene_amd import render as _rmod
_orig_pp = _rmod.post_process
def _pp(*x, **k):
    _mark("render: between blend and maps (radii > 0, python)")
    r = _orig_pp(*x, **k)
    _mark("render: pp kernels")
    return r
_rmod.post_process = _pp
rasterizer._prepare, rasterizer.rasterize_gaussians = _prep, _rg
_orig_opt = tr.opt.step
def _opt(*x, **k):
    if a.delay_at == "opt": _spin(a.delay_us)
    r = _orig_opt(*x, **k); _mark("adam"); return r
tr.opt.step = _opt
_blocked = [0.0, 0]
_orig_verify = rasterizer._verify_pending
def _timed_verify(owner):
    t = time.perf_counter()
    _orig_verify(owner)
    _blocked[0] += time.perf_counter() - t
    _blocked[1] += 1
rasterizer._verify_pending = _timed_verify
for it in range(5):
    tr.step(it)
_blocked[0] = 0.0
_ev.clear(); _host_t.clear()
_e0 = torch.cuda.Event(enable_timing=True); _e0.record(); torch.cuda.synchronize(); _h0 = time.perf_counter()
torch.cuda.synchronize()
ms0 = torch.cuda.memory_stats()
t0 = time.perf_counter()
for it in range(5, 5 + a.steps):
    tr.step(it)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step, total %.3f ms/step, tail after last enqueue %.3f ms" %
      (1e3 * (t1 - t0) / a.steps, 1e3 * (t2 - t0) / a.steps, 1e3 * (t2 - t1)))
print("host blocked on the binning-size event: %.3f ms/step" % (1e3 * _blocked[0] / a.steps))
print("host WORK per step (enqueue minus the time blocked on that event - the host runs ahead of the GPU and is throttled by "
      "it there): %.3f ms; the iteration behind the blend goes through %s" %
      (1e3 * (t1 - t0) / a.steps - 1e3 * _blocked[0] / a.steps,
       "one C entry (isr_seg_step_tail)" if os.environ.get("ISR_C_TAIL", "1") != "0" else "the autograd graph (ISR_C_TAIL=0)"))
ms1 = torch.cuda.memory_stats()
for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "allocation.all.allocated", "segment.all.allocated"):
    print(k, ms1.get(k, 0) - ms0.get(k, 0))
print("reserved GB %.2f allocated peak GB %.2f" % (ms1["reserved_bytes.all.current"] / 2**30, ms1["allocated_bytes.all.peak"] / 2**30))
if os.environ.get("HOST_CPROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for it in range(5 + a.steps, 5 + 2 * a.steps):
        tr.step(it)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
    sys.exit(0)
acc = {}
for (t0_, e0), (t1_, e1) in zip(_ev[:-1], _ev[1:]):
    key = t1_ if t1_ != "start" else "between steps"
    acc[key] = acc.get(key, 0.0) + e0.elapsed_time(e1)
print("GPU phases (ms/step):", {k: round(v / a.steps, 3) for k, v in acc.items()})
# host lead: how long before the GPU reaches a mark was it enqueued? (one step in the middle of the run)
starts = [i for i, (t, _) in enumerate(_ev) if t == "start"]
lo, hi = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
print("mark                                                   host(ms)   gpu(ms)   host lead(ms)")
for i in range(lo, hi + 1):
    tag, e = _ev[i]
    g = _e0.elapsed_time(e)
    h = 1e3 * (_host_t[i] - _h0)
    print("%-52s %9.3f %9.3f %9.3f" % (tag, h, g, g - h))
