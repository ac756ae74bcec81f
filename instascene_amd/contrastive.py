"""``contrastive_loss`` — host-side mirror of the reference's
``utils/contrastive_utils.contrastive_loss`` (utils/contrastive_utils.py:18-73):
same signature, same label filtering / dense relabelling (:25-50, torch ops, as
in the reference), with the arithmetic core (:41-71) running in the HIP library
(``iso_contrastive_forward/backward``: exact-fp32 MFMA similarity, atomic-free
deterministic reductions)."""
from __future__ import annotations

import ctypes

import torch

from ._lib import check, lib


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _ProtoNCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, labels, predef_u, K, temp_lambda):
        L = lib()
        feats = features.contiguous().float()
        N, F = feats.shape
        labels = labels.to(torch.int32).contiguous()
        pre = predef_u.contiguous().float() if predef_u is not None else None
        nbytes = L.iso_contrastive_scratch_bytes(N, F, K)
        state = torch.empty(nbytes, dtype=torch.uint8, device=feats.device)
        loss = torch.empty(1, dtype=torch.float32, device=feats.device)
        with torch.cuda.device(feats.device):
            check(L.iso_contrastive_forward(N, F, K, _p(feats), _p(labels), _p(pre), float(temp_lambda), _p(loss),
                                            _p(state), nbytes, _stream()), "iso_contrastive_forward")
        ctx.save_for_backward(labels, state, pre if pre is not None else torch.empty(0, device=feats.device))
        ctx.dims = (N, F, K, nbytes, pre is not None)
        return loss[0]

    @staticmethod
    def backward(ctx, grad_loss):
        L = lib()
        labels, state, pre = ctx.saved_tensors
        N, F, K, nbytes, has_pre = ctx.dims
        g = grad_loss.reshape(1).contiguous().float()
        out = torch.empty((N, F), dtype=torch.float32, device=labels.device)
        with torch.cuda.device(labels.device):
            check(L.iso_contrastive_backward(N, F, K, _p(labels), _p(pre) if has_pre else None, _p(g), _p(out),
                                             _p(state), nbytes, _stream()), "iso_contrastive_backward")
        return out, None, None, None, None


def contrastive_loss(features, masks, predef_u_list=None, min_pixnum=0, temp_lambda=1000, consider_negative=False):
    """Drop-in for ``utils.contrastive_utils.contrastive_loss`` (reference :18-73).

    features [N,F] float (CUDA), masks [N] integer labels; returns the summed ProtoNCE loss."""
    if not features.is_cuda:
        raise RuntimeError("contrastive_loss: features must be a CUDA tensor (the HIP library is the only backend)")
    if not consider_negative:
        valid = masks > 0
    else:
        valid = torch.ones_like(masks, dtype=torch.bool)
    mask_ids, mask_nums = torch.unique(masks, return_counts=True)
    valid_mask_ids = mask_ids[mask_nums > min_pixnum]
    valid = valid & torch.isin(masks, valid_mask_ids)
    labels = masks[valid].to(torch.int64)
    if not consider_negative:
        labels = labels - 1
    feats = features[valid, :]
    present = torch.unique(labels)                  # sorted, like the reference's remapping (:43-50)
    K = int(present.numel())
    if K == 0:
        return feats.sum() * 0.0
    remap = torch.zeros(int(present.max().item()) + 1, dtype=torch.long, device=labels.device)
    remap[present] = torch.arange(K, device=labels.device)
    dense = remap[labels]
    u = predef_u_list[present] if predef_u_list is not None else None
    return _ProtoNCE.apply(feats, dense, u, K, float(temp_lambda))
