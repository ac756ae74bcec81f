"""ORACLE — TEST INFRASTRUCTURE ONLY.

Plain-PyTorch (CPU) restatements of the torch-level ops on the hot path, each
pinned against fixtures produced by the reference's own Python
(tests/golden/make_goldens.py):

* ``contrastive_loss``  — utils/contrastive_utils.py:18-73   (golden: contrastive_loss.npz)
* ``render_post``       — gaussian_renderer/__init__.py:118-169 (golden: render_post.npz)
* ``depth_to_normal``   — utils/point_utils.py:10-40          (golden: depth_to_normal.npz)
* ``l1`` / ``ssim``     — utils/loss_utils.py:18-83           (golden: losses.npz)
* ``gram_schmidt``      — scene/gaussian_model.py:161-167     (golden: gram_schmidt.npz)

Written from the algorithm (dense one-hot algebra instead of the reference's
scatter/unique plumbing) so that they are an independent check of the product's
fused kernels on arbitrary seeded inputs where no golden exists.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as Fnn


def contrastive_loss(features, labels, predef_u=None, min_pixnum=0, temp_lambda=1000, consider_negative=False):
    """ProtoNCE-style loss.  features [N,F], labels [N] int.  Returns scalar (sum)."""
    labels = labels.to(torch.int64)
    keep = torch.ones_like(labels, dtype=torch.bool) if consider_negative else labels > 0
    ids, counts = torch.unique(labels, return_counts=True)
    big = ids[counts > min_pixnum]
    keep = keep & (labels.unsqueeze(1) == big.unsqueeze(0)).any(dim=1)
    lab = labels[keep]
    if not consider_negative:
        lab = lab - 1
    f = features[keep]
    f = f / (f.norm(dim=-1, keepdim=True) + 1e-9).detach()
    present = torch.unique(lab)                                   # sorted
    onehot = (lab.unsqueeze(1) == present.unsqueeze(0)).to(f.dtype)   # [N,K]
    n_k = onehot.sum(0)                                            # [K]
    if predef_u is not None:
        U = predef_u[present]
    else:
        U = (onehot.t() @ f) / n_k.unsqueeze(1)
    diff_norm = (f - onehot @ U).norm(dim=1)                       # [N]
    phi = (onehot.t() @ diff_norm) / (n_k * torch.log(n_k + temp_lambda))
    phi = torch.clamp(phi * 10, min=0.5, max=1.0).detach()
    sim = torch.exp((f @ U.t()) / phi.unsqueeze(0))                # [N,K]
    pos = (sim * onehot).sum(1)
    return -(torch.log(pos / (sim.sum(1) + 1e-9))).sum()


def depths_to_points(wvt, full_proj, W, H, depth):
    """utils/point_utils.py:10-27; wvt/full_proj in the reference's row-vector storage."""
    c2w = wvt.t().inverse()
    n2p = torch.tensor([[W / 2, 0, 0, W / 2], [0, H / 2, 0, H / 2], [0, 0, 0, 1]], dtype=wvt.dtype,
                       device=depth.device).t()
    intr = ((c2w.t() @ full_proj) @ n2p)[:3, :3].t()
    gx, gy = torch.meshgrid(torch.arange(W, device=depth.device).to(wvt.dtype),
                            torch.arange(H, device=depth.device).to(wvt.dtype), indexing="xy")
    pts = torch.stack([gx, gy, torch.ones_like(gx)], dim=-1).reshape(-1, 3)
    rays = pts @ intr.inverse().t() @ c2w[:3, :3].t()
    return depth.reshape(-1, 1) * rays + c2w[:3, 3]


def depth_to_normal(wvt, full_proj, W, H, depth):
    pts = depths_to_points(wvt, full_proj, W, H, depth).reshape(H, W, 3)
    out = torch.zeros_like(pts)
    dx = pts[2:, 1:-1] - pts[:-2, 1:-1]
    dy = pts[1:-1, 2:] - pts[1:-1, :-2]
    out[1:-1, 1:-1] = Fnn.normalize(torch.linalg.cross(dx, dy, dim=-1), dim=-1)
    return out


def render_post(allmap, wvt, full_proj, W, H, depth_ratio):
    """Post-processing of the 7-channel map (gaussian_renderer/__init__.py:127-167)."""
    alpha = allmap[1:2]
    n = allmap[2:5]
    n = (n.permute(1, 2, 0) @ wvt[:3, :3].t()).permute(2, 0, 1)
    med = torch.nan_to_num(allmap[5:6], 0, 0)
    exp_d = torch.nan_to_num(allmap[0:1] / alpha, 0, 0)
    dist = allmap[6:7]
    surf_depth = exp_d * (1 - depth_ratio) + depth_ratio * med
    surf_normal = depth_to_normal(wvt, full_proj, W, H, surf_depth).permute(2, 0, 1) * alpha.detach()
    return dict(rend_alpha=alpha, rend_normal=n, rend_dist=dist, surf_depth=surf_depth, surf_normal=surf_normal,
                rend_depth=exp_d, rend_median_depth=med)


def l1(a, b):
    return (a - b).abs().mean()


def _window(size=11, sigma=1.5, channels=3, dtype=torch.float32):
    g = torch.tensor([math.exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(size)])
    g = (g / g.sum()).unsqueeze(1)
    w2 = (g @ g.t()).float()
    return w2.expand(channels, 1, size, size).contiguous().to(dtype)


def ssim(a, b, size=11):
    ch = a.shape[-3]
    w = _window(size, 1.5, ch, a.dtype).to(a.device)
    conv = lambda x: Fnn.conv2d(x, w, padding=size // 2, groups=ch)
    mu1, mu2 = conv(a), conv(b)
    s1 = conv(a * a) - mu1 * mu1
    s2 = conv(b * b) - mu2 * mu2
    s12 = conv(a * b) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2))).mean()


def gram_schmidt(vectors):
    out = []
    for v in vectors:
        for u in out:
            v = v - torch.dot(v, u) * u
        out.append(v / (v.norm() + 1e-9))
    return torch.stack(out)
