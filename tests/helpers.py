"""Shared helpers for the parity tests (seeded scenes, oracle calls, comparisons)."""
import math

import numpy as np
import torch

import oracle
from instascene_amd import scenes


def small_scene(P=400, F=6, W=64, H=48, seed=3, mu_s=math.log(0.06), ncam=6, sh_degree=3):
    sc = scenes.synthetic_scene(P, F, seed, mu_s)
    cams = scenes.ring_cameras(ncam, W, H)
    inp = scenes.activated_inputs(sc)
    return sc, cams, inp


def oracle_forward(inp, cam, bg=(0.0, 0.0, 0.0), sh_degree=3, tracer=False, scale_modifier=1.0, fma=False, margins=False, **over):
    a = {k: (None if v is None else v.detach().cpu().numpy()) for k, v in inp.items()}
    a.update(over)
    return oracle.forward(a["means3D"], a["opacities"], cam.world_view_transform.numpy(),
                          cam.full_proj_transform.numpy(), cam.camera_center.numpy(),
                          np.asarray(bg, np.float32), cam.image_width, cam.image_height,
                          math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), scales=a.get("scales"),
                          rotations=a.get("rotations"), shs=a.get("shs"), colors_precomp=a.get("colors_precomp"),
                          transMat_precomp=a.get("transMat_precomp"), extra=a.get("extra"), sh_degree=sh_degree,
                          scale_modifier=scale_modifier, tracer=tracer, fma=fma, margins=margins)


def rel_err(a, b, eps=1e-12):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + eps))


def assert_close(a, b, rtol, name="", atol_scale=1.0, atol=0.0):
    """max|a-b| <= rtol * max|b| + atol — the tolerance form used for images/gradients.

    ``atol`` is only used for the distortion map, whose fp32 evaluation
    (m^2 A + M2 - 2 m M1 with m ~ 1) cancels catastrophically: its absolute
    fp32 noise floor is ~1e-7 per contributor regardless of its own magnitude."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{name}: shape {a.shape} vs {b.shape}"
    scale = np.abs(b).max()
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= rtol * scale * atol_scale + atol + 1e-30, f"{name}: max err {err:.3e} > {rtol:g} * {scale:.3e}"


def row_rel_errors(a, b, floor=1e-3):
    """Per-row relative error |a_row - b_row|_max / max(|b_row|_max, floor * |b|_max) - the max-normalised form of
    :func:`assert_close` lets a row whose magnitude is 1e-4 of the tensor's maximum be 100 % wrong; this one does not (rows
    below ``floor`` of the maximum are measured against that floor: fp32 sums of thousands of pixel terms carry an absolute
    noise of ~1e-7 of the largest term, which no relative bound survives on a row that is itself that small)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    a = a.reshape(a.shape[0], -1)
    b = b.reshape(a.shape)
    scale = np.abs(b).max() + 1e-300
    row = np.maximum(np.abs(b).max(axis=1), floor * scale)
    return np.abs(a - b).max(axis=1) / row


def assert_rows_close(a, b, name="", q=99.9, rtol=1e-2, floor=1e-3):
    """The q-th percentile of the per-row relative error (see :func:`row_rel_errors`) is at most ``rtol``."""
    if np.asarray(b).size == 0:
        return 0.0
    e = row_rel_errors(a, b, floor)
    v = float(np.percentile(e, q))
    assert v <= rtol, f"{name}: {q}th percentile of the per-row relative error is {v:.3e} > {rtol:g} (max {e.max():.3e})"
    return v
