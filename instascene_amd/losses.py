"""Image losses of the reference's ``train.py`` step (utils/loss_utils.py:18-83), restated in
torch (conv2d plumbing) so that the RGB harness exercises the full geometry backward.  Pinned against the
reference's own outputs in tests/golden/losses.npz."""
from __future__ import annotations

import torch
import torch.nn.functional as F

_WINDOWS = {}


def l1_loss(network_output, gt):
    return torch.abs(network_output - gt).mean()


def cos_loss(network_output, gt):
    return (1 - (network_output * gt).sum(dim=0)).mean()


def _taps(size: int, device, dtype):
    """Normalised 1-D Gaussian taps (sigma 1.5).  The reference's 2-D window is the outer product of these
    (utils/loss_utils.py:28-36), so filtering rows then columns is the same linear operator at 2*size instead of
    size**2 multiplies per pixel."""
    key = (size, str(device), dtype)
    t = _WINDOWS.get(key)
    if t is None:
        x = torch.arange(size, dtype=torch.float64) - size // 2
        t = torch.exp(-x * x / (2.0 * 1.5 * 1.5))
        t = (t / t.sum()).to(device=device, dtype=dtype)
        _WINDOWS[key] = t
    return t


def _blur(planes, taps):
    """Zero-padded separable blur of every plane of a [B, C, H, W] stack (depthwise)."""
    c, k = planes.size(1), taps.numel()
    rows = F.conv2d(planes, taps.view(1, 1, 1, k).expand(c, 1, 1, k), padding=(0, k // 2), groups=c)
    return F.conv2d(rows, taps.view(1, 1, k, 1).expand(c, 1, k, 1), padding=(k // 2, 0), groups=c)


class _SsimHip(torch.autograd.Function):
    """Mean SSIM of two [C,H,W] CUDA images: ``iso_ssim_forward/backward`` (separable window through LDS).  The gradient
    flows to the first image only (the second is the ground truth in train.py:91)."""

    @staticmethod
    def forward(ctx, img1, img2):
        from ._lib import check, lib
        L = lib()
        a, b = img1.contiguous().float(), img2.detach().contiguous().float()
        C, H, W = a.shape
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        need = img1.requires_grad
        dmaps = torch.empty((3, C, H, W), dtype=torch.float32, device=a.device) if need else None
        nbytes = L.iso_ssim_scratch_bytes(C, H, W)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
        with torch.cuda.device(a.device):
            check(L.iso_ssim_forward(C, H, W, _ptr(a), _ptr(b), _ptr(out), _ptr(dmaps), _ptr(scratch), nbytes, _stream()),
                  "iso_ssim_forward")
        ctx.save_for_backward(a, b, dmaps)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        from ._lib import check, lib
        a, b, dmaps = ctx.saved_tensors
        if dmaps is None:
            return None, None
        C, H, W = a.shape
        gm = g.reshape(1).contiguous().float()
        out = torch.empty_like(a)
        with torch.cuda.device(a.device):
            check(lib().iso_ssim_backward(C, H, W, _ptr(a), _ptr(b), _ptr(dmaps), _ptr(gm), _ptr(out), _stream()),
                  "iso_ssim_backward")
        return out, None


class _PhotometricHip(torch.autograd.Function):
    """``(1 - lambda) * L1 + lambda * (1 - SSIM)`` of two [C,H,W] CUDA images (train.py:89-91) in the SSIM kernels
    (``iso_photometric_forward/backward``): the images are in LDS anyway, so the L1 mean and its sign gradient cost no pass.
    Returns ``(loss, l1, ssim)``; gradients flow to the first image."""

    @staticmethod
    def forward(ctx, img1, img2, lambda_dssim):
        from ._lib import check, lib
        L = lib()
        a, b = img1.contiguous().float(), img2.detach().contiguous().float()
        C, H, W = a.shape
        out = torch.empty(2, dtype=torch.float32, device=a.device)       # ssim mean, l1 mean
        need = img1.requires_grad
        dmaps = torch.empty((3, C, H, W), dtype=torch.float32, device=a.device) if need else None
        nbytes = L.iso_ssim_scratch_bytes(C, H, W)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
        with torch.cuda.device(a.device):
            check(L.iso_photometric_forward(C, H, W, _ptr(a), _ptr(b), _ptr(out[0:1]), _ptr(out[1:2]), _ptr(dmaps),
                                            _ptr(scratch), nbytes, _stream()), "iso_photometric_forward")
        ctx.save_for_backward(a, b, dmaps)
        ctx.lam = float(lambda_dssim)
        ssim_v, l1_v = out[0], out[1]
        loss = (1.0 - ctx.lam) * l1_v + ctx.lam * (1.0 - ssim_v)
        ctx.mark_non_differentiable(l1_v, ssim_v)
        return loss, l1_v, ssim_v

    @staticmethod
    def backward(ctx, g, _g_l1, _g_ssim):
        from ._lib import check, lib
        a, b, dmaps = ctx.saved_tensors
        if dmaps is None or g is None:
            return None, None, None
        C, H, W = a.shape
        gg = g.reshape(1).float()
        gs = torch.stack(((-ctx.lam) * gg[0], (1.0 - ctx.lam) * gg[0]))   # dL/d ssim_mean, dL/d l1_mean
        out = torch.empty_like(a)
        with torch.cuda.device(a.device):
            check(lib().iso_photometric_backward(C, H, W, _ptr(a), _ptr(b), _ptr(dmaps), _ptr(gs[0:1]), _ptr(gs[1:2]),
                                                 _ptr(out), _stream()), "iso_photometric_backward")
        return out, None, None


def photometric_loss(image, gt, lambda_dssim: float = 0.2):
    """``(1 - lambda_dssim) * l1_loss + lambda_dssim * (1 - ssim)`` (train.py:89-91).  CUDA [C,H,W] images: one fused pair of
    kernels; otherwise composed from :func:`l1_loss` and :func:`ssim`."""
    if image.is_cuda and image.dim() == 3 and gt.shape == image.shape and not gt.requires_grad:
        return _PhotometricHip.apply(image, gt, float(lambda_dssim))[0]
    return (1.0 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1.0 - ssim(image, gt))


class _TrainLossHip(torch.autograd.Function):
    """The loss of one ``train.py`` iteration (train.py:89-103) - photometric term, distortion and normal-consistency
    regularisers - in three launches forward and one backward (``iso_train_loss_forward/backward``) instead of ~25 torch
    kernels around the SSIM pair.  Returns ``(total, parts[5])`` with ``parts = total, L1, SSIM, normal error, distortion``
    (means); gradients flow to ``image``, ``rend_normal``, ``surf_normal`` and ``rend_dist``."""

    @staticmethod
    def forward(ctx, image, gt, lambda_dssim, rend_normal, surf_normal, lambda_normal, rend_dist, lambda_dist):
        from ._lib import check, lib
        L = lib()
        a, b = image.contiguous().float(), gt.detach().contiguous().float()
        C, H, W = a.shape
        use_n = rend_normal is not None and surf_normal is not None and float(lambda_normal) != 0.0
        use_d = rend_dist is not None and float(lambda_dist) != 0.0
        rn = rend_normal.contiguous().float() if use_n else None
        sn = surf_normal.contiguous().float() if use_n else None
        rd = rend_dist.contiguous().float() if use_d else None
        if use_n and (tuple(rn.shape) != (3, H, W) or tuple(sn.shape) != (3, H, W)):
            raise ValueError("train_loss: rend_normal / surf_normal must be [3,H,W] of the image's size")
        if use_d and rd.numel() != H * W:
            raise ValueError("train_loss: rend_dist must hold H*W values")
        out = torch.empty(5, dtype=torch.float32, device=a.device)
        dmaps = torch.empty((3, C, H, W), dtype=torch.float32, device=a.device)
        nbytes = L.iso_train_loss_scratch_bytes(C, H, W)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
        with torch.cuda.device(a.device):
            check(L.iso_train_loss_forward(C, H, W, _ptr(a), _ptr(b), float(lambda_dssim), _ptr(rn), _ptr(sn),
                                           float(lambda_normal), _ptr(rd), float(lambda_dist), _ptr(out), _ptr(dmaps),
                                           _ptr(scratch), nbytes, _stream()), "iso_train_loss_forward")
        ctx.save_for_backward(a, b, dmaps, rn, sn)
        ctx.cfg = (float(lambda_dssim), float(lambda_normal) if use_n else 0.0, float(lambda_dist) if use_d else 0.0,
                   None if rend_dist is None else tuple(rend_dist.shape), use_d)
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)
        return out[0], out

    @staticmethod
    def backward(ctx, g, _g_parts):
        from ._lib import check, lib
        a, b, dmaps, rn, sn = ctx.saved_tensors
        if g is None:
            return (None,) * 8
        lam, ln, ldist, dist_shape, use_d = ctx.cfg
        C, H, W = a.shape
        gg = g.reshape(1).contiguous().float()
        d_img = torch.empty_like(a)
        # the library produces both normal gradients or neither; each is handed back only where autograd asked for it
        want_n = rn is not None and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        d_rn = torch.empty_like(rn) if want_n else None
        d_sn = torch.empty_like(sn) if want_n else None
        d_rd = torch.empty(dist_shape, dtype=torch.float32, device=a.device) if (use_d and ctx.needs_input_grad[6]) else None
        with torch.cuda.device(a.device):
            check(lib().iso_train_loss_backward(C, H, W, _ptr(a), _ptr(b), _ptr(dmaps), lam, _ptr(rn), _ptr(sn), ln, ldist,
                                                _ptr(gg), _ptr(d_img), _ptr(d_rn), _ptr(d_sn), _ptr(d_rd), _stream()),
                  "iso_train_loss_backward")
        return (d_img, None, None, d_rn if ctx.needs_input_grad[3] else None, d_sn if ctx.needs_input_grad[4] else None, None,
                d_rd, None)


def train_loss(image, gt, lambda_dssim, rend_normal=None, surf_normal=None, lambda_normal=0.0, rend_dist=None,
               lambda_dist=0.0):
    """``(1 - l) L1 + l (1 - SSIM) + lambda_dist * rend_dist.mean() + lambda_normal * (1 - (rend_normal * surf_normal).sum(0)).mean()``
    - the total loss of train.py:89-103.  CUDA images: ``iso_train_loss_forward/backward``; otherwise composed in torch."""
    if image.is_cuda and image.dim() == 3 and gt.shape == image.shape and not gt.requires_grad:
        return _TrainLossHip.apply(image, gt, float(lambda_dssim), rend_normal, surf_normal, float(lambda_normal), rend_dist,
                                   float(lambda_dist))[0]
    loss = (1.0 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1.0 - ssim(image, gt))
    if rend_dist is not None and lambda_dist != 0.0:
        loss = loss + lambda_dist * rend_dist.mean()
    if rend_normal is not None and lambda_normal != 0.0:
        loss = loss + lambda_normal * (1 - (rend_normal * surf_normal).sum(dim=0))[None].mean()
    return loss


def _ptr(t):
    import ctypes
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    import ctypes
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ssim(img1, img2, window_size: int = 11, size_average: bool = True):
    """Structural similarity with an 11x11 Gaussian window (utils/loss_utils.py:39-63).  [C,H,W] CUDA images with the
    default window and ``size_average=True`` (what train.py uses) run in the HIP library; otherwise the five local
    moments (E[a], E[b], E[a^2], E[b^2], E[ab]) are filtered as one stacked depthwise torch pass (this restatement is
    what tests/golden/losses.npz pins on the host)."""
    if (img1.is_cuda and img1.dim() == 3 and img2.shape == img1.shape and window_size == 11 and size_average
            and not img2.requires_grad):
        return _SsimHip.apply(img1, img2)
    a = img1 if img1.dim() == 4 else img1.unsqueeze(0)
    b = img2 if img2.dim() == 4 else img2.unsqueeze(0)
    c = a.size(1)
    taps = _taps(window_size, a.device, a.dtype)
    m = _blur(torch.cat((a, b, a * a, b * b, a * b), dim=1), taps)
    ea, eb, eaa, ebb, eab = m.split(c, dim=1)
    k1, k2 = 0.01 ** 2, 0.03 ** 2
    cov = eab - ea * eb
    var_sum = (eaa - ea * ea) + (ebb - eb * eb)
    mean_sq_sum = ea * ea + eb * eb
    quality = (2 * ea * eb + k1) * (2 * cov + k2) / ((mean_sq_sum + k1) * (var_sum + k2))
    if size_average:
        return quality.mean()
    return quality.flatten(1).mean(1)
