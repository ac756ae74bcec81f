"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference hot path (InstaScene's 2DGS surfel rasterizer,
contrastive loss and 3-NN initialiser).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker / reported CPU baseline.  The product
package ``instascene_amd`` never imports it.

* ``surfel_oracle.cpp``  — C++17/OpenMP restatement of
  ``cuda_rasterizer/{forward,backward,rasterizer_impl}.cu`` and simple-knn
  (each routine cites reference file:line).
* ``torch_surfel.py``    — independent differentiable PyTorch restatement of the
  forward blend (autograd gradients cross-check the hand-derived backward).
* ``torch_ops.py``       — torch restatements of ``contrastive_loss``,
  ``render()`` post-processing, ``depth_to_normal``, L1/SSIM.

Pinning status is documented in the header of ``surfel_oracle.cpp`` and in
DESIGN.md ("parity unpinned" for the per-pixel CUDA loops: the reference ships
no golden vectors and cannot be built here).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_float, c_int, c_int32, c_int64, c_uint8, c_uint32, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libsurfel_oracle.so")
_LIB_FMA_PATH = os.path.join(_HERE, "_build", "libsurfel_oracle_fma.so")
_lib = None
_lib_fma = None


def build(force: bool = False) -> str:
    """Compile the C++ oracle with g++ (a few seconds)."""
    src = os.path.join(_HERE, "surfel_oracle.cpp")
    if force or any(not os.path.exists(q) or os.path.getmtime(q) < os.path.getmtime(src) for q in (_LIB_PATH, _LIB_FMA_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib_fma():
    """The second build of the same source (``-ffp-contract=fast``, libm ``expf``): what a compiler with the reference's
    defaults (nvcc ``--fmad=true``, libdevice ``expf``) may legitimately produce.  Only the per-pixel loops are taken from it
    (``forward(..., fma=True)``): K1 and the binning stay the primary build's, so both see the same tile lists."""
    global _lib_fma
    if _lib_fma is None:
        build()
        _lib_fma = ctypes.CDLL(_LIB_FMA_PATH)
    return _lib_fma


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.so_bin.restype = c_int64
        _lib.so_num_threads.restype = c_int
    return _lib


NATIVE_FLAGS = "-O3 -march=native -std=c++17 -fPIC -shared -fopenmp -ffp-contract=off -fno-fast-math"
BUILD_FLAGS = "-O2 -std=c++17 -fPIC -shared -fopenmp -mfma -ffp-contract=off -fno-fast-math"
_native = None


def use_native_build() -> str:
    """bench.py's cpu_baseline only: rebuild the same source ON THIS MACHINE with the flags SURVEY 8(d) names (-O3 -march=native;
    contraction still off, so the results keep their bits) and route forward / backward through it.  The default build (-O2,
    portable: it travels prebuilt to another box) stays what the tests check against.  Returns the flags in use."""
    global _lib, _native
    if _native is not None:
        return _native
    import hashlib
    try:
        cpu = [l for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        cpu = "unknown"
    out = os.path.join(_HERE, "_build", "libsurfel_oracle_native_%s.so" % hashlib.sha1(cpu.encode()).hexdigest()[:8])
    src = os.path.join(_HERE, "surfel_oracle.cpp")
    try:
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            os.makedirs(os.path.dirname(out), exist_ok=True)
            subprocess.check_call([os.environ.get("CXX", "g++")] + NATIVE_FLAGS.split() + [src, "-o", out])
        nat = ctypes.CDLL(out)
        nat.so_bin.restype = c_int64
        nat.so_num_threads.restype = c_int
        _lib, _native = nat, NATIVE_FLAGS
    except Exception:
        lib()
        _native = BUILD_FLAGS + " (the -O3 -march=native build failed on this machine)"
    return _native


def _p(a, ty=None):
    if a is None:
        return None
    return a.ctypes.data_as(c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def num_threads() -> int:
    return int(lib().so_num_threads())


def mark_visible(means3D, view, proj):
    means3D = _f32(means3D)
    P = means3D.shape[0]
    out = np.zeros(P, np.uint8)
    lib().so_mark_visible(c_int(P), _p(means3D), _p(_f32(view)), _p(_f32(proj)), _p(out))
    return out.astype(bool)


def dist2_3nn(points):
    points = _f32(points)
    P = points.shape[0]
    out = np.zeros(P, np.float32)
    lib().so_dist2_3nn(c_int(P), _p(points), _p(out))
    return out


def forward(means3D, opacities, view, proj, campos, bg, W, H, tanfovx, tanfovy, *, scales=None,
            rotations=None, shs=None, colors_precomp=None, transMat_precomp=None, extra=None,
            sh_degree=0, scale_modifier=1.0, tracer=False, fma=False, margins=False):
    """Full forward (K1..K8).  Returns a dict with outputs and every intermediate
    the backward / parity tests need.  Argument meaning follows
    ``CudaRasterizer::Rasterizer::forward`` (rasterizer_impl.cu:198-351)."""
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    opacities = _f32(opacities).reshape(-1)
    scales, rotations, shs = _f32(scales), _f32(rotations), _f32(shs)
    colors_precomp, transMat_precomp, extra = _f32(colors_precomp), _f32(transMat_precomp), _f32(extra)
    view, proj, campos, bg = _f32(view).reshape(-1), _f32(proj).reshape(-1), _f32(campos), _f32(bg)
    ED = 0 if extra is None or extra.size == 0 else extra.shape[1]
    M = 0 if shs is None else shs.shape[1]
    N = W * H
    gx, gy = (W + 15) // 16, (H + 15) // 16

    radii = np.zeros(P, np.int32)
    means2D = np.zeros((P, 2), np.float32)
    depths = np.zeros(P, np.float32)
    transMats = np.zeros((P, 9), np.float32)
    rgb = np.zeros((P, 3), np.float32)
    normal_opacity = np.zeros((P, 4), np.float32)
    tiles_touched = np.zeros(P, np.uint32)
    clamped = np.zeros((P, 3), np.uint8)
    L.so_preprocess_fwd(c_int(P), c_int(sh_degree), c_int(M), _p(means3D), _p(scales), c_float(scale_modifier),
                        _p(rotations), _p(opacities), _p(shs), _p(transMat_precomp), _p(colors_precomp),
                        _p(view), _p(proj), _p(campos), c_int(W), c_int(H), _p(radii), _p(means2D), _p(depths),
                        _p(transMats), _p(rgb), _p(normal_opacity), _p(tiles_touched), _p(clamped))
    R = int(L.so_bin(c_int(P), c_int(W), c_int(H), _p(radii), _p(means2D), _p(depths), _p(tiles_touched),
                     None, None, None))
    keys = np.zeros(max(R, 1), np.uint64)
    point_list = np.zeros(max(R, 1), np.uint32)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    L.so_bin(c_int(P), c_int(W), c_int(H), _p(radii), _p(means2D), _p(depths), _p(tiles_touched), _p(keys),
             _p(point_list), _p(ranges))
    keys, point_list = keys[:R], point_list[:R]

    colors_used = colors_precomp if colors_precomp is not None else rgb
    tm_used = transMat_precomp if transMat_precomp is not None else transMats
    final_T = np.zeros((3, N), np.float32)
    n_contrib = np.zeros((2, N), np.uint32)
    out_color = np.zeros((3, H, W), np.float32)
    out_others = np.zeros((7, H, W), np.float32)
    out_extra = np.zeros((max(ED, 0), H, W), np.float32)
    trace = np.full((N * 10, 2), -1, np.int32) if tracer else None
    tcount = c_int64(0)
    pl = point_list if R > 0 else np.zeros(1, np.uint32)
    mg = np.zeros((5, N), np.float32) if margins else None
    (lib_fma() if fma else L).so_render_fwd_margins(
        c_int(W), c_int(H), c_int(ED), _p(ranges), _p(pl), _p(means2D), _p(colors_used), _p(tm_used),
        _p(extra), _p(normal_opacity), _p(bg), _p(final_T), _p(n_contrib), _p(out_color),
        _p(out_others), _p(out_extra) if ED else None, _p(trace), c_int64(N * 10), ctypes.byref(tcount), _p(mg))
    st = dict(P=P, W=W, H=H, ED=ED, M=M, R=R, sh_degree=sh_degree, scale_modifier=scale_modifier,
              tanfovx=tanfovx, tanfovy=tanfovy, radii=radii, means2D=means2D, depths=depths,
              transMats=transMats, rgb=rgb, normal_opacity=normal_opacity, tiles_touched=tiles_touched,
              clamped=clamped, keys=keys, point_list=point_list, ranges=ranges, final_T=final_T,
              n_contrib=n_contrib, color=out_color, others=out_others, extra=out_extra,
              colors_used=colors_used, tm_used=tm_used,
              inputs=dict(means3D=means3D, opacities=opacities, scales=scales, rotations=rotations, shs=shs,
                          colors_precomp=colors_precomp, transMat_precomp=transMat_precomp, extra=extra,
                          view=view, proj=proj, campos=campos, bg=bg))
    if tracer:
        st["tracer"] = trace[: int(tcount.value)]
    if margins:
        st["margins"] = mg
    return st


def backward(st, dL_dcolor, dL_dothers, dL_dextra=None):
    """K9 + K10 (rasterizer_impl.cu:355-463).  Returns the nine gradient arrays of
    ``RasterizeGaussiansBackwardCUDA`` (rasterize_points.cu:261) plus dL_dnormal."""
    L = lib()
    P, W, H, ED, M = st["P"], st["W"], st["H"], st["ED"], st["M"]
    inp = st["inputs"]
    dL_dcolor, dL_dothers = _f32(dL_dcolor), _f32(dL_dothers)
    dL_dextra = _f32(dL_dextra) if ED else None
    g = dict(dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
             dL_dopacity=np.zeros((P, 1), np.float32), dL_dmeans3D=np.zeros((P, 3), np.float32),
             dL_dtransMat=np.zeros((P, 9), np.float32), dL_dsh=np.zeros((P, M, 3), np.float32),
             dL_dscales=np.zeros((P, 2), np.float32), dL_drotations=np.zeros((P, 4), np.float32),
             dL_dextra=np.zeros((P, max(ED, 0)), np.float32), dL_dnormal=np.zeros((P, 3), np.float32))
    pl = st["point_list"] if st["R"] > 0 else np.zeros(1, np.uint32)
    L.so_render_bwd(c_int(W), c_int(H), c_int(ED), c_int(P), _p(st["ranges"]), _p(pl), _p(inp["bg"]),
                    _p(st["means2D"]), _p(st["normal_opacity"]), _p(st["tm_used"]), _p(st["colors_used"]),
                    _p(inp["extra"]), _p(st["final_T"]), _p(st["n_contrib"]), _p(dL_dcolor), _p(dL_dothers),
                    _p(dL_dextra), _p(g["dL_dtransMat"]), _p(g["dL_dmeans2D"]), _p(g["dL_dnormal"]),
                    _p(g["dL_dopacity"]), _p(g["dL_dcolors"]), _p(g["dL_dextra"]) if ED else None)
    g["raw_dL_dmeans2D"] = g["dL_dmeans2D"].copy()      # before the densification overwrite
    g["raw_dL_dtransMat"] = g["dL_dtransMat"].copy()
    focal_y = np.float32(H) / (np.float32(2.0) * np.float32(st["tanfovy"]))
    focal_x = np.float32(W) / (np.float32(2.0) * np.float32(st["tanfovx"]))
    L.so_preprocess_bwd(c_int(P), c_int(st["sh_degree"]), c_int(M), _p(inp["means3D"]), _p(st["tm_used"]),
                        _p(st["radii"]), _p(inp["shs"]), _p(st["clamped"]), _p(inp["scales"]),
                        _p(inp["rotations"]), c_float(st["scale_modifier"]), _p(inp["view"]), _p(inp["proj"]),
                        c_float(focal_x), c_float(focal_y), c_float(st["tanfovx"]), c_float(st["tanfovy"]),
                        _p(inp["campos"]), _p(g["dL_dtransMat"]), _p(g["dL_dnormal"]), _p(g["dL_dcolors"]),
                        _p(g["dL_dsh"]), _p(g["dL_dmeans2D"]), _p(g["dL_dmeans3D"]), _p(g["dL_dscales"]),
                        _p(g["dL_drotations"]))
    return g


def preprocess_backward(st, dL_dtransMat, dL_dnormal=None, dL_dcolors=None, dL_dmeans2D=None):
    """K10 alone (backward.cu:469-656) on the state of :func:`forward`: per-Gaussian gradients
    ``(dL_dmeans3D, dL_dscales, dL_drotations, dL_dsh)`` from given ``dL/dtransMat [P,9]`` (+ optional ``dL/dnormal [P,3]``,
    ``dL/dcolor [P,3]``, ``dL/dmean2D [P,3]``) - what tests/golden/kten_backward.npz pins against the reference's autograd."""
    L = lib()
    P, M = st["P"], st["M"]
    inp = st["inputs"]
    z = lambda *s: np.zeros(s, np.float32)
    g = dict(dL_dtransMat=_f32(dL_dtransMat).copy(), dL_dnormal=z(P, 3) if dL_dnormal is None else _f32(dL_dnormal).copy(),
             dL_dcolors=z(P, 3) if dL_dcolors is None else _f32(dL_dcolors).copy(), dL_dsh=z(P, M, 3),
             dL_dmeans2D=z(P, 3) if dL_dmeans2D is None else _f32(dL_dmeans2D).copy(), dL_dmeans3D=z(P, 3),
             dL_dscales=z(P, 2), dL_drotations=z(P, 4))
    focal_y = np.float32(st["H"]) / (np.float32(2.0) * np.float32(st["tanfovy"]))
    focal_x = np.float32(st["W"]) / (np.float32(2.0) * np.float32(st["tanfovx"]))
    L.so_preprocess_bwd(c_int(P), c_int(st["sh_degree"]), c_int(M), _p(inp["means3D"]), _p(st["tm_used"]),
                        _p(st["radii"]), _p(inp["shs"]), _p(st["clamped"]), _p(inp["scales"]),
                        _p(inp["rotations"]), c_float(st["scale_modifier"]), _p(inp["view"]), _p(inp["proj"]),
                        c_float(focal_x), c_float(focal_y), c_float(st["tanfovx"]), c_float(st["tanfovy"]),
                        _p(inp["campos"]), _p(g["dL_dtransMat"]), _p(g["dL_dnormal"]), _p(g["dL_dcolors"]),
                        _p(g["dL_dsh"]), _p(g["dL_dmeans2D"]), _p(g["dL_dmeans3D"]), _p(g["dL_dscales"]),
                        _p(g["dL_drotations"]))
    return g


def test_quat_to_rot(q):
    q = _f32(q)
    out = np.zeros((q.shape[0], 3, 3), np.float32)
    lib().so_test_quat_to_rot(c_int(q.shape[0]), _p(q), _p(out))
    return out


def test_sh_to_rgb(deg, pos, cam, shs):
    pos, cam, shs = _f32(pos), _f32(cam), _f32(shs)
    n = pos.shape[0]
    rgb = np.zeros((n, 3), np.float32)
    cl = np.zeros((n, 3), np.uint8)
    lib().so_test_sh_to_rgb(c_int(n), c_int(deg), _p(pos), _p(cam), _p(shs), _p(rgb), _p(cl))
    return rgb, cl.astype(bool)


def test_tile_rect(cx, cy, r, gx, gy):
    out = np.zeros(4, np.uint32)
    lib().so_test_tile_rect(c_float(cx), c_float(cy), c_int(r), c_int(gx), c_int(gy), _p(out))
    return tuple(int(v) for v in out)


def test_exp(x):
    x = _f32(x)
    y = np.zeros_like(x)
    lib().so_test_exp(c_int(x.size), _p(x), _p(y))
    return y
