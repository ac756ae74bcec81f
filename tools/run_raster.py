#!/usr/bin/env python
"""Run the rasterizer forward (+ feature-only backward) on a BASELINE config a few times — a small,
torch-free-ish driver for rocprofv3 (kernel trace / PMC passes)."""
import argparse, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instascene_amd import scenes, rasterizer as rz
from instascene_amd._lib import GRAD_EXTRA, GRAD_GEOMETRY, MODE_EXACT, MODE_FAST

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--mode", default="fast")
ap.add_argument("--geom", type=int, default=0)
ap.add_argument("--sparse", type=int, default=1, help="dL/dfeature nonzero only at 16384 sampled pixels (train_semantic)")
a = ap.parse_args()
mode = MODE_FAST if a.mode == "fast" else MODE_EXACT
scene, cams, cfg = scenes.config_scene(a.config)
inp = {k: (v.cuda() if v is not None else None) for k, v in scenes.activated_inputs(scene).items()}
e = torch.empty(0, device="cuda")
W, H, F = cfg["W"], cfg["H"], cfg["F"]
for it in range(a.iters):
    cam = cams[it % len(cams)]
    args = (torch.zeros(3, device="cuda"), inp["means3D"], e, inp["opacities"], inp["scales"], inp["rotations"], 1.0, e,
            inp["extra"] if F else e, F, cam.world_view_transform.cuda(), cam.full_proj_transform.cuda(),
            math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), H, W, inp["shs"], 3, cam.camera_center.cuda(), False, False)
    torch.cuda.synchronize(); t0 = time.time()
    out = rz.rasterize_gaussians(*args, mode=mode, tracer=False)
    torch.cuda.synchronize(); t1 = time.time()
    R, color, others, radii, extra, geom, binning, img = out[:8]
    dC = torch.zeros_like(color) if not a.geom else torch.randn_like(color)
    dO = torch.zeros_like(others) if not a.geom else torch.randn_like(others)
    if F:
        if a.sparse == 2:
            dE = torch.zeros_like(extra)
        elif a.sparse:
            dE = torch.zeros_like(extra)
            idx = torch.randint(0, W * H, (16384,), device="cuda")
            dE.view(F, -1)[:, idx] = torch.randn(F, 16384, device="cuda")
        else:
            dE = torch.randn_like(extra)
    else:
        dE = e
    mask = (GRAD_EXTRA if F else 0) | (GRAD_GEOMETRY if a.geom else 0)
    torch.cuda.synchronize(); t2 = time.time()
    g = rz.rasterize_gaussians_backward(args[0], args[1], radii, e, args[4], args[5], args[8], 1.0, e, args[10], args[11],
                                        args[12], args[13], dC, dO, dE, args[16], 3, args[18], geom, R, binning, img,
                                        False, grad_mask=mask, mode=mode)
    torch.cuda.synchronize(); t3 = time.time()
    import ctypes
    from instascene_amd._lib import lib
    L = lib(); L.isr_profile_enable(1)
    g = rz.rasterize_gaussians_backward(args[0], args[1], radii, e, args[4], args[5], args[8], 1.0, e, args[10], args[11],
                                        args[12], args[13], dC, dO, dE, args[16], 3, args[18], geom, R, binning, img,
                                        False, grad_mask=mask, mode=mode)
    buf = ctypes.create_string_buffer(1 << 14); L.isr_profile_summary(buf, len(buf)); L.isr_profile_enable(0)
    print(f"iter {it}: R={R} fwd {1e3*(t1-t0):.2f} ms  bwd {1e3*(t3-t2):.2f} ms | " + buf.value.decode().replace("\n", "; "))
