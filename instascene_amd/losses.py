"""Image losses of the reference's ``train.py`` step (utils/loss_utils.py:18-83), restated in
torch (conv2d plumbing) so that the RGB harness exercises the full geometry backward.  Pinned against the
reference's own outputs in tests/golden/losses.npz."""
from __future__ import annotations

from math import exp

import torch
import torch.nn.functional as F

_WINDOWS = {}


def l1_loss(network_output, gt):
    return torch.abs(network_output - gt).mean()


def cos_loss(network_output, gt):
    return (1 - (network_output * gt).sum(dim=0)).mean()


def _window(size: int, channel: int, device, dtype):
    key = (size, channel, str(device), dtype)
    w = _WINDOWS.get(key)
    if w is None:
        g = torch.tensor([exp(-(x - size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(size)])
        g = (g / g.sum()).unsqueeze(1)
        w = (g @ g.t()).float().unsqueeze(0).unsqueeze(0).expand(channel, 1, size, size).contiguous()
        w = w.to(device=device, dtype=dtype)
        _WINDOWS[key] = w
    return w


def ssim(img1, img2, window_size: int = 11, size_average: bool = True):
    channel = img1.size(-3)
    w = _window(window_size, channel, img1.device, img1.dtype)
    pad = window_size // 2
    mu1 = F.conv2d(img1, w, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, w, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, w, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, w, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, w, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)
