"""``distCUDA2`` — mirror of ``simple_knn._C.distCUDA2`` (submodules/simple-knn/spatial.cu:15-26):
mean squared distance to the three nearest neighbours of every point, on the HIP library."""
from __future__ import annotations

import ctypes

import torch

from ._lib import check, lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: points must be a CUDA tensor")
    pts = points.contiguous().float()
    P = pts.shape[0]
    out = torch.zeros((P,), dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    L = lib()
    nbytes = L.iso_knn_scratch_bytes(P)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        check(L.iso_dist2_3nn(P, ctypes.c_void_p(pts.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                              ctypes.c_void_p(scratch.data_ptr()), nbytes,
                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "iso_dist2_3nn")
    return out
