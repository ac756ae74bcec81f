"""Device-side counterpart of ``get_segmap_gaussians`` (spatial_track/modules/init_tracker.py:16-47), the
only consumer of the rasterizer's ``gau_related_pixels`` tracer.  The reference
loops over mask ids in Python and builds ``set(tensor.tolist())`` for each; here the (mask, gaussian)
pairs are deduplicated with one sort on the device and split per mask."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch


def segmap_gaussians(gau_related_pixels: torch.Tensor, segmap: torch.Tensor, min_gaussians: int = 50
                     ) -> Tuple[Dict[int, torch.Tensor], torch.Tensor]:
    """gau_related_pixels [K,2] int (gaussian id, pixel id); segmap [H,W] (or flat) integer mask ids.

    Returns ``(mask_info, frame_gaussian_ids)``: ``mask_info[mask_id]`` = sorted unique ids of the Gaussians
    traced into that mask (masks with fewer than ``min_gaussians`` distinct Gaussians and mask 0 are dropped,
    reference :37-44); ``frame_gaussian_ids`` = sorted unique ids of every traced Gaussian (:35)."""
    g = gau_related_pixels[:, 0].to(torch.int64)
    pix = gau_related_pixels[:, 1].to(torch.int64)
    frame_ids = torch.unique(g)
    labels = segmap.reshape(-1).to(torch.int64)[pix]
    keep = labels != 0
    g, labels = g[keep], labels[keep]
    if g.numel() == 0:
        return {}, frame_ids
    span = int(g.max().item()) + 1
    pairs = torch.unique(labels * span + g)                  # sorted by (label, gaussian)
    plab = torch.div(pairs, span, rounding_mode="floor")
    pg = pairs - plab * span
    ulab, counts = torch.unique_consecutive(plab, return_counts=True)
    ends = torch.cumsum(counts, 0)
    starts = ends - counts
    out: Dict[int, torch.Tensor] = {}
    for lab, s, e, c in zip(ulab.tolist(), starts.tolist(), ends.tolist(), counts.tolist()):
        if c >= min_gaussians:
            out[int(lab)] = pg[s:e]
    return out, frame_ids


def get_segmap_gaussians(gaussian, view, render_fn=None, background=None, min_gaussians: int = 50):
    """Same call shape as the reference function: renders ``view`` and reduces its tracer list."""
    from .render import render
    render_fn = render if render_fn is None else render_fn
    dev = gaussian.get_xyz.device
    if background is None:
        background = torch.zeros(3, dtype=torch.float32, device=dev)
    with torch.no_grad():
        pkg = render_fn(view, gaussian, gaussian.pipelineparams, background)
    mask_info, frame_ids = segmap_gaussians(pkg["gau_related_pixels"], view.segmap.to(dev), min_gaussians)
    return {k: set(v.tolist()) for k, v in mask_info.items()}, frame_ids.tolist()
