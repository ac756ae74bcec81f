"""The C++ oracle's forward and hand-derived backward vs the independent
differentiable torch restatement (autograd) — pins K9/K10 maths without the
reference binary."""
import math

import numpy as np
import pytest
import torch

import oracle
from oracle import torch_surfel as ts
from helpers import small_scene, oracle_forward, assert_close


def _torch_forward(inp, cam, st, bg, dt=torch.float64, leaves=None):
    W, H = cam.image_width, cam.image_height
    L = leaves
    T, normal, centre, rgb, _ = ts.per_gaussian(L["means3D"], L["scales"], L["rotations"], L["shs"],
                                                cam.world_view_transform, cam.full_proj_transform,
                                                cam.camera_center, W, H, 3)
    return ts.blend(T, normal, centre, L["opacities"].reshape(-1), rgb, L["extra"], torch.tensor(bg), W, H,
                    st["ranges"], st["point_list"])


@pytest.mark.parametrize("seed,bg", [(3, (0.0, 0.0, 0.0)), (11, (0.3, 0.6, 0.1))])
def test_forward_and_gradients_match_autograd(seed, bg):
    # 64x48: K10 re-derives W,H as int(focal*tan*2) (backward.cu:633-634); exact for this size
    sc, cams, inp = small_scene(P=300, F=5, W=64, H=48, seed=seed)
    cam = cams[1]
    st = oracle_forward(inp, cam, bg=bg)
    assert st["R"] > 0
    leaves = {k: v.detach().to(torch.float64).requires_grad_(True) for k, v in inp.items()}
    color, others, extra = _torch_forward(inp, cam, st, bg, leaves=leaves)
    assert_close(st["color"], color.detach().numpy(), 1e-4, "color")
    assert_close(st["extra"], extra.detach().numpy(), 1e-4, "extra")
    for ch, nm in enumerate(["depth", "alpha", "nx", "ny", "nz", "median", "dist"]):
        assert_close(st["others"][ch], others[ch].detach().numpy(), 1e-4, nm, atol=2e-6 if nm == "dist" else 0.0)

    g = torch.Generator().manual_seed(seed)
    dC = torch.randn(color.shape, generator=g, dtype=torch.float64)
    dO = torch.randn(others.shape, generator=g, dtype=torch.float64)
    dE = torch.randn(extra.shape, generator=g, dtype=torch.float64)
    loss = (color * dC).sum() + (others * dO).sum() + (extra * dE).sum()
    loss.backward()
    grads = oracle.backward(st, dC.numpy(), dO.numpy(), dE.numpy())
    vis = st["radii"] > 0
    tol = 2e-4
    assert_close(grads["dL_dextra"], leaves["extra"].grad.numpy(), tol, "dL_dextra")
    assert_close(grads["dL_dopacity"], leaves["opacities"].grad.numpy(), tol, "dL_dopacity")
    assert_close(grads["dL_dsh"], leaves["shs"].grad.numpy(), tol, "dL_dsh")
    assert_close(grads["dL_dscales"], leaves["scales"].grad.numpy(), tol, "dL_dscales")
    assert_close(grads["dL_drotations"], leaves["rotations"].grad.numpy(), tol, "dL_drot")
    assert_close(grads["dL_dmeans3D"], leaves["means3D"].grad.numpy(), tol, "dL_dmeans3D")
    assert vis.sum() > 50


def test_raw_blend_gradients_match_autograd():
    """K9 alone: gradients w.r.t. the per-Gaussian blend inputs (T, centre, normal, opacity, rgb, extra)."""
    sc, cams, inp = small_scene(P=250, F=4, W=40, H=40, seed=5)
    cam = cams[2]
    bg = (0.2, 0.1, 0.7)
    st = oracle_forward(inp, cam, bg=bg)
    dt = torch.float64
    T = torch.tensor(st["transMats"], dtype=dt).reshape(-1, 3, 3).requires_grad_(True)
    centre = torch.tensor(st["means2D"], dtype=dt).requires_grad_(True)
    no = torch.tensor(st["normal_opacity"], dtype=dt)
    normal = no[:, :3].clone().requires_grad_(True)
    opac = no[:, 3].clone().requires_grad_(True)
    rgb = torch.tensor(st["rgb"], dtype=dt).requires_grad_(True)
    extra = inp["extra"].to(dt).requires_grad_(True)
    color, others, ex = ts.blend(T, normal, centre, opac, rgb, extra, torch.tensor(bg), 40, 40, st["ranges"],
                                 st["point_list"])
    g = torch.Generator().manual_seed(0)
    dC = torch.randn(color.shape, generator=g, dtype=dt)
    dO = torch.randn(others.shape, generator=g, dtype=dt)
    dE = torch.randn(ex.shape, generator=g, dtype=dt)
    ((color * dC).sum() + (others * dO).sum() + (ex * dE).sum()).backward()
    grads = oracle.backward(st, dC.numpy(), dO.numpy(), dE.numpy())
    tol = 2e-4
    assert_close(grads["raw_dL_dtransMat"], T.grad.reshape(-1, 9).numpy(), tol, "dL_dT")
    assert_close(grads["raw_dL_dmeans2D"][:, :2], centre.grad.numpy(), tol, "dL_dcentre")
    assert_close(grads["dL_dnormal"], normal.grad.numpy(), tol, "dL_dnormal")
    assert_close(grads["dL_dopacity"][:, 0], opac.grad.numpy(), tol, "dL_dopacity")
    assert_close(grads["dL_dcolors"], rgb.grad.numpy(), tol, "dL_dcolors")
    assert_close(grads["dL_dextra"], extra.grad.numpy(), tol, "dL_dextra")
