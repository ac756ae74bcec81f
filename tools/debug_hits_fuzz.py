#!/usr/bin/env python
"""k_pack_hits' masks on the fuzz scenes of tests/test_gpu_fuzz.py (extreme anisotropy, opacities on the 1/255 threshold):
isr_debug_check_hit_masks per scene; for an offending (Gaussian, pixel) the splat's record and its conic in float64."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import test_gpu_fuzz as FZ  # noqa: E402
import test_gpu_rasterizer as T  # noqa: E402
from instascene_amd._lib import lib  # noqa: E402

L = lib()
cases = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(40))
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = 0
for case in cases:
    inp, cam, F = FZ._scene(case, seed0)
    for mode in (T.MODE_EXACT, T.MODE_FAST):
        args, out = T.hip_forward(inp, cam, mode=mode)
        R, geom, binning, img = out[0], out[5], out[6], out[7]
        P = inp["means3D"].shape[0]
        H, W = out[1].shape[1], out[1].shape[2]
        chk = torch.zeros(8, dtype=torch.int64, device="cuda")
        rc = L.isr_debug_check_hit_masks(P, W, H, int(R), ctypes.c_void_p(geom.data_ptr()), ctypes.c_void_p(binning.data_ptr()),
                                         ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(chk.data_ptr()),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        c = [int(v) for v in chk.tolist()]
        rec = {"case": case, "mode": int(mode), "R": int(R), "clear": c[0], "fast_pass": c[1], "exact_pass": c[2], "set": c[3], "set_idle": c[4]}
        if c[1] or c[2]:
            bad += 1
            gid = c[5] - 1
            px, py = c[6] & 0xffffffff, c[6] >> 32
            dbg = T.rz.debug_state(P, W, H, int(R), geom, binning, img)
            rec.update({"gaussian": gid, "pixel": [px, py], "opacity": float(inp["opacities"][gid]), "scale": inp["scales"][gid].tolist()})
            if dbg is not None:
                r = dbg["records"][gid].astype(np.float64)
                rec["record"] = [float(v) for v in r]
        print(json.dumps(rec))
print("offending (case, mode):", bad)
