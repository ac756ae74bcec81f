"""How many walk iterations would the forward blend need if each 8x4 half / 4x4 quad of an 8x8 block walked its own hit sub-list?
(STATS build of k_render_fwd_fast_w, counters 8..15 - include/instascene_rasterizer.h.)  Usage: python tools/sublist_stats.py [C3]"""
import ctypes, json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instascene_amd import scenes, rasterizer as rz
from instascene_amd._lib import MODE_FAST, lib

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C3"
scene, cams, cfg = scenes.config_scene(cfgname)
inp = {k: (v.cuda() if v is not None else None) for k, v in scenes.activated_inputs(scene).items()}
e = torch.empty(0, device="cuda")
W, H, F = cfg["W"], cfg["H"], cfg["F"]
for view in range(2):
    cam = cams[view]
    args = (torch.zeros(3, device="cuda"), inp["means3D"], e, inp["opacities"], inp["scales"], inp["rotations"], 1.0, e,
            inp["extra"] if F else e, F, cam.world_view_transform.cuda(), cam.full_proj_transform.cuda(),
            math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), H, W, inp["shs"], 3, cam.camera_center.cuda(), False, False)
    c = torch.zeros(16, dtype=torch.int64, device="cuda")
    lib().isr_forward_set_counters(ctypes.c_void_p(c.data_ptr()))
    out = rz.rasterize_gaussians(*args, mode=MODE_FAST, tracer=False)
    torch.cuda.synchronize()
    v = c.tolist()
    ev, bl = v[1], v[2]
    print(json.dumps({"config": cfgname, "view": view, "R": int(out[0]), "evaluated": ev, "blending": bl, "lane_pairs": v[3],
                      "near(any lane inside band.hi)": {"half_walk": v[8], "quad_walk": v[9], "half_entries": v[10], "quad_entries": v[11],
                                                         "half_walk/evaluated": round(v[8] / ev, 4), "quad_walk/evaluated": round(v[9] / ev, 4)},
                      "blending": {"half_walk": v[12], "quad_walk": v[13], "half_entries": v[14], "quad_entries": v[15],
                                   "half_walk/blending": round(v[12] / bl, 4), "quad_walk/blending": round(v[13] / bl, 4)}}), flush=True)
