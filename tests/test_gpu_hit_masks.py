"""k_pack_hits' per-(tile entry, block half) hit masks (csrc/isr_binning.hip) against the blend kernels' own per-pixel tests.

A clear bit removes a (block half, splat) pair from what the forward blend, the EXACT forward and the splat-major geometry backward
walk; it is sound iff no pixel of the half would have blended the splat - in FAST arithmetic (fast_pair_lane: FAST's decision,
EXACT's inside the guard bands) and in EXACT arithmetic (the reference's tests, forward.cu:356-393).  The device-side checker
(isr_debug_check_hit_masks) evaluates every cleared pair on all 32 pixels and counts such pixels: the count must be 0.  The masks
are also meant to be tight: of the halves left set, few hold no blending pixel (the bounding-octagon test of rounds 1-5 left a third).
"""
import ctypes
import math

import numpy as np
import pytest
import torch

import test_gpu_fuzz as FZ
import test_gpu_rasterizer as T
from instascene_amd import scenes
from instascene_amd._lib import lib

pytestmark = pytest.mark.gpu


def _check(out, P, W, H):
    R, geom, binning, img = out[0], out[5], out[6], out[7]
    chk = torch.zeros(8, dtype=torch.int64, device="cuda")
    rc = lib().isr_debug_check_hit_masks(P, W, H, int(R), ctypes.c_void_p(geom.data_ptr()), ctypes.c_void_p(binning.data_ptr()),
                                         ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(chk.data_ptr()),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    return [int(v) for v in chk.tolist()]


@pytest.mark.parametrize("seed0", [1000, 7000])
def test_hit_masks_hide_no_blending_pair_on_the_fuzz_scenes(seed0):
    """40 scenes each: needles of aspect 300 seen thousands of pixels from their centre, 0.3-pixel splats, opacities on the 1/255
    threshold, horizon-crossing surfels - in both arithmetic modes."""
    cleared = 0
    for case in range(40):
        inp, cam, F = FZ._scene(case, seed0)
        for mode in (T.MODE_EXACT, T.MODE_FAST):
            args, out = T.hip_forward(inp, cam, mode=mode)
            c = _check(out, inp["means3D"].shape[0], out[1].shape[2], out[1].shape[1])
            assert c[1] == 0 and c[2] == 0, (case, mode, c, "Gaussian %d at pixel (%d, %d)" % (c[5] - 1, c[6] & 0xffffffff, c[6] >> 32))
            cleared += c[0]
    assert cleared > 0


@pytest.mark.parametrize("config,idle_max", [("C1", 0.08), ("C3", 0.08)])
def test_hit_masks_at_full_size_are_sound_and_tight(config, idle_max):
    scene, cams, cfg = scenes.config_scene(config)
    inp = {k: (v.cuda() if v is not None else None) for k, v in scenes.activated_inputs(scene).items()}
    e = torch.empty(0, device="cuda")
    W, H, F = cfg["W"], cfg["H"], cfg["F"]
    for v in (0, 5):
        cam = cams[v]
        args = (torch.zeros(3, device="cuda"), inp["means3D"], e, inp["opacities"], inp["scales"], inp["rotations"], 1.0, e,
                inp["extra"] if F else e, F, cam.world_view_transform.cuda(), cam.full_proj_transform.cuda(),
                math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), H, W, inp["shs"], 3, cam.camera_center.cuda(), False, False)
        out = T.rz.rasterize_gaussians(*args, mode=T.MODE_FAST, tracer=False)
        c = _check(out, cfg["P"], W, H)
        assert c[1] == 0 and c[2] == 0, (config, v, c)
        # tight: halves left set that hold no blending pixel (octagon alone: 0.35)
        assert c[4] <= idle_max * c[3], (config, v, c[4] / max(1, c[3]))
        assert c[0] > c[3]          # most (entry, half) pairs of the reference's 3-sigma rectangles are cleared
