#!/usr/bin/env python
"""A/B timing of the forward blend kernel alone: `ISR_LIB_PATH=<variant .so> python tools/fwd_ab.py [--config C3] [--mode fast]`
renders 8 ring views 3 times each and prints the kernel's mean time from HIP events (isr_profile), plus a checksum of the
outputs so that variants can be compared for equality."""
import argparse, ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instascene_amd import scenes, rasterizer as rz
from instascene_amd._lib import MODE_EXACT, MODE_FAST, lib, LIB_PATH

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--mode", default="fast")
ap.add_argument("--tracer", type=int, default=1)
ap.add_argument("--feat", type=int, default=1)
a = ap.parse_args()
mode = MODE_FAST if a.mode == "fast" else MODE_EXACT
scene, cams, cfg = scenes.config_scene(a.config)
scene = scenes.spatially_sorted(scene)
inp = {k: (v.cuda() if v is not None else None) for k, v in scenes.activated_inputs(scene).items()}
e = torch.empty(0, device="cuda")
W, H, F = cfg["W"], cfg["H"], (cfg["F"] if a.feat else 0)
L = lib()
tot, n, chk = 0.0, 0, 0.0
for rep in range(4):
    for vi in range(8):
        cam = cams[vi * 2]
        args = (torch.zeros(3, device="cuda"), inp["means3D"], e, inp["opacities"], inp["scales"], inp["rotations"], 1.0, e,
                inp["extra"] if F else e, F, cam.world_view_transform.cuda(), cam.full_proj_transform.cuda(),
                math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), H, W, inp["shs"], 3, cam.camera_center.cuda(), False, False)
        if rep == 1 and vi == 0:
            torch.cuda.synchronize()
            L.isr_profile_enable(1)
        out = rz.rasterize_gaussians(*args, mode=mode, tracer=bool(a.tracer))
        if rep == 0:
            chk += float(out[1].double().sum()) + float(out[2].double().sum()) + (float(out[4].double().sum()) if F else 0.0)
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 14)
L.isr_profile_summary(buf, len(buf))
L.isr_profile_enable(0)
line = {l.split()[0]: float(l.split()[2]) / int(l.split()[1]) for l in buf.value.decode().splitlines()}
print(f"{os.path.basename(LIB_PATH)} {a.config} {a.mode} F={F}: " + "  ".join(f"{k} {v:.4f}" for k, v in line.items()) + f"  checksum {chk:.6f}")
