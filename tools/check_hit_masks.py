#!/usr/bin/env python
"""k_pack_hits' hit masks of full-size views against the blend kernels' own per-pixel tests (isr_debug_check_hit_masks) and what
the masks cost the kernels: the forward's evaluation counters and per-kernel times.  One JSON line per (config, view).

    python tools/check_hit_masks.py [--configs C2,C3,C5] [--views 2] [--mode fast_reflists]
    ISR_PACK_EXACT=0 python tools/check_hit_masks.py ...      # the bounding-octagon masks of rounds 1-5, for comparison
"""
import argparse
import ctypes
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from instascene_amd import rasterizer as rz, scenes  # noqa: E402
from instascene_amd._lib import GRAD_GEOMETRY, MODE_EXACT, MODE_FAST, lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="C2,C3")
ap.add_argument("--views", type=int, default=2)
ap.add_argument("--mode", default="fast_reflists")
a = ap.parse_args()
rz.set_mode(a.mode)
L = lib()
e = torch.empty(0, device="cuda")
for config in a.configs.split(","):
    scene, cams, cfg = scenes.config_scene(config)
    inp = {k: (v.cuda() if v is not None else None) for k, v in scenes.activated_inputs(scene).items()}
    W, H, F = cfg["W"], cfg["H"], cfg["F"]
    for v in range(a.views):
        cam = cams[(7 * v) % len(cams)]
        args = (torch.zeros(3, device="cuda"), inp["means3D"], e, inp["opacities"], inp["scales"], inp["rotations"], 1.0, e,
                inp["extra"] if F else e, F, cam.world_view_transform.cuda(), cam.full_proj_transform.cuda(),
                math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), H, W, inp["shs"], 3, cam.camera_center.cuda(), False, False)
        mode = MODE_EXACT if a.mode == "exact" else MODE_FAST
        rz.rasterize_gaussians(*args, mode=mode, tracer=False)            # warm-up
        fwd = torch.zeros(16, dtype=torch.int64, device="cuda")
        if mode == MODE_FAST:
            L.isr_forward_set_counters(ctypes.c_void_p(fwd.data_ptr()))
        out = rz.rasterize_gaussians(*args, mode=mode, tracer=False)
        R, color, others, radii, extra, geom, binning, img = out[:8]
        chk = torch.zeros(8, dtype=torch.int64, device="cuda")
        rc = L.isr_debug_check_hit_masks(inp["means3D"].shape[0], W, H, int(R), ctypes.c_void_p(geom.data_ptr()),
                                         ctypes.c_void_p(binning.data_ptr()), ctypes.c_void_p(img.data_ptr()),
                                         ctypes.c_void_p(chk.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        c = [int(x) for x in chk.tolist()]
        f = [int(x) for x in fwd.tolist()]
        # kernel times: forward + dense geometry backward, HIP events on the launch stream
        L.isr_profile_enable(1)
        for _ in range(3):
            out = rz.rasterize_gaussians(*args, mode=mode, tracer=False)
            R, color, others, radii, extra, geom, binning, img = out[:8]
            if not F:
                rz.rasterize_gaussians_backward(args[0], args[1], radii, e, args[4], args[5], args[8], 1.0, e, args[10], args[11],
                                                args[12], args[13], torch.ones_like(color), torch.ones_like(others), e, args[16], 3,
                                                args[18], geom, R, binning, img, False, grad_mask=GRAD_GEOMETRY, mode=mode)
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 14)
        L.isr_profile_summary(buf, len(buf))
        L.isr_profile_enable(0)
        times = {}
        for line in buf.value.decode().splitlines():
            name, cnt, tot = line.split()
            times[name] = round(float(tot) / int(cnt), 4)
        print(json.dumps({"config": config, "view": v, "mode": a.mode, "pack_exact": os.environ.get("ISR_PACK_EXACT", "1") != "0", "R": int(R),
                          "halves_with_a_clear_bit": c[0], "FAST_near_pixels_in_them": c[1], "EXACT_passing_pixels_in_them": c[2],
                          "halves_with_a_set_bit": c[3], "of_them_with_no_near_pixel": c[4],
                          "forward_wave_splat_evaluations": f[1], "forward_evaluations_that_blend": f[2], "forward_blending_lanes": f[3],
                          "ms": {k: times[k] for k in ("k_render_fwd", "k_render_bwd", "k_pack_hits", "k_preprocess") if k in times}}))
