#!/usr/bin/env python
"""Host cost of one forward + backward through the two bindings of the C ABI: the ctypes layer (instascene_amd.rasterizer: arena,
asynchronous binning, prefetch) and the compiled torch extension (instascene_amd._C_hip = diff_surfel_rasterization._C: the
reference's blocking semantics).  A tiny scene, so that the GPU is never the limit: wall time per call = host time."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from instascene_amd import scenes, rasterizer as rz, _C_hip
from instascene_amd._lib import GRAD_EXTRA, GRAD_GEOMETRY, MODE_FAST

P, F, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 2000, 16, 128, 96
sc = scenes.synthetic_scene(P, F, 3, math.log(0.06))
cam = scenes.ring_cameras(4, W, H)[0]
inp = {k: (v.cuda() if v is not None else None) for k, v in scenes.activated_inputs(sc).items()}
e = torch.empty(0, device="cuda")
a = (torch.zeros(3, device="cuda"), inp["means3D"], e, inp["opacities"], inp["scales"], inp["rotations"], 1.0, e, inp["extra"], F,
     cam.world_view_transform.cuda(), cam.full_proj_transform.cuda(), math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), H, W, inp["shs"], 3,
     cam.camera_center.cuda(), False, False)
dC, dO, dE = torch.randn(3, H, W).cuda(), torch.randn(7, H, W).cuda(), torch.randn(F, H, W).cuda()
rz.set_mode("fast_reflists")


def ctypes_layer():
    o = rz.rasterize_gaussians(*a, tracer=True)
    rz.rasterize_gaussians_backward(a[0], a[1], o[3], e, a[4], a[5], a[8], 1.0, e, a[10], a[11], a[12], a[13], dC, dO, dE, a[16], 3, a[18],
                                    o[5], o[0], o[6], o[7], False, grad_mask=GRAD_EXTRA | GRAD_GEOMETRY, mode=MODE_FAST)


def compiled():
    o = _C_hip.rasterize_gaussians(*a)
    _C_hip.rasterize_gaussians_backward(a[0], a[1], o[3], e, a[4], a[5], a[8], 1.0, e, a[10], a[11], a[12], a[13], dC, dO, dE, a[16], 3, a[18],
                                        o[5], o[0], o[6], o[7], False)


for name, fn in (("ctypes layer (instascene_amd.rasterizer)", ctypes_layer), ("compiled extension (instascene_amd._C_hip)", compiled)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: {1e6 * (t1 - t0) / n:.1f} us per forward + backward on the host ({1e6 * (t2 - t0) / n:.1f} us incl. the GPU's tail), P = {P}, {W}x{H}, F = {F}")
