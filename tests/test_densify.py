"""Adaptive density control against the reference's own outputs: tests/golden/densify.npz was written by
tests/golden/make_goldens.py running the reference's GaussianModel.add_densification_stats / densify_and_prune /
reset_opacity on the host with seeded inputs and a seeded RNG stream."""
import os

import numpy as np
import pytest
import torch

from instascene_amd.densify import Densifier, GROUPS, rotation_matrices

NAMES = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
         "rotation": "_rotation"}
LR = {"xyz": 0.00016, "f_dc": 0.0025, "f_rest": 0.0025 / 20.0, "opacity": 0.05, "scaling": 0.005, "rotation": 0.001}


class _Model:
    pass


def _build(z, device="cpu"):
    m = _Model()
    for n, attr in NAMES.items():
        setattr(m, attr, torch.nn.Parameter(torch.tensor(z["init_" + n]).to(device).requires_grad_(True)))
    opt = torch.optim.Adam([{"params": [getattr(m, NAMES[n])], "lr": LR[n], "name": n} for n in GROUPS], lr=0.0, eps=1e-15)
    for s in range(2):
        for n in GROUPS:
            getattr(m, NAMES[n]).grad = torch.tensor(z[f"grad{s}_{n}"]).to(device)
        opt.step()
        opt.zero_grad(set_to_none=True)
    return m, opt


def test_rotation_matrices_match_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "sh_rot.npz"))
    got = rotation_matrices(torch.tensor(z["quats"]))
    np.testing.assert_allclose(got.numpy(), z["rotmats"], rtol=0, atol=1e-6)


def test_densification_matches_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "densify.npz"))
    m, opt = _build(z)
    d = Densifier(m, opt, percent_dense=float(z["params"][4]))
    for s in range(3):
        d.accumulate(torch.tensor(z[f"vsgrad{s}"]), torch.tensor(z[f"vis{s}"]), torch.tensor(z[f"radii{s}"]))
    np.testing.assert_array_equal(d.xyz_gradient_accum.numpy(), z["stats_accum"])
    np.testing.assert_array_equal(d.denom.numpy(), z["stats_denom"])
    np.testing.assert_array_equal(d.max_radii2D.numpy(), z["stats_max_radii"])
    torch.manual_seed(int(z["seed"]))
    max_grad, min_opacity, extent, max_screen = (float(v) for v in z["params"][:4])
    d.densify_and_prune(max_grad, min_opacity, extent, max_screen)
    assert m._xyz.shape[0] == z["after_xyz"].shape[0]
    for n in GROUPS:
        p = getattr(m, NAMES[n])
        assert p.requires_grad and opt.param_groups[GROUPS.index(n)]["params"][0] is p
        np.testing.assert_allclose(p.detach().numpy(), z["after_" + n], rtol=0, atol=2e-6, err_msg=n)
        st = opt.state[p]
        np.testing.assert_array_equal(st["exp_avg"].numpy(), z["after_m_" + n])
        np.testing.assert_array_equal(st["exp_avg_sq"].numpy(), z["after_v_" + n])
    np.testing.assert_array_equal(d.xyz_gradient_accum.numpy(), z["after_accum"])
    np.testing.assert_array_equal(d.denom.numpy(), z["after_denom"])
    np.testing.assert_array_equal(d.max_radii2D.numpy(), z["after_max_radii"])
    d.reset_opacity()
    np.testing.assert_allclose(m._opacity.detach().numpy(), z["reset_opacity"], rtol=0, atol=1e-6)
    st = opt.state[m._opacity]
    np.testing.assert_array_equal(st["exp_avg"].numpy(), z["reset_m"])
    np.testing.assert_array_equal(st["exp_avg_sq"].numpy(), z["reset_v"])
    # the optimiser still steps on the edited groups
    for n in GROUPS:
        getattr(m, NAMES[n]).grad = torch.ones_like(getattr(m, NAMES[n]))
    opt.step()


@pytest.mark.gpu
def test_densify_statistics_kernel_matches_host_path(golden_dir):
    """iso_densify_stats (one pass) against the masked torch ops, and the whole densify/prune on device tensors: same
    row counts and the same deterministic parts as the golden (the split offsets use the device RNG)."""
    z = np.load(os.path.join(golden_dir, "densify.npz"))
    m, opt = _build(z, "cuda")
    d = Densifier(m, opt, percent_dense=float(z["params"][4]))
    for s in range(3):
        d.accumulate(torch.tensor(z[f"vsgrad{s}"]).cuda(), torch.tensor(z[f"vis{s}"]).cuda(), torch.tensor(z[f"radii{s}"]).cuda())
    np.testing.assert_allclose(d.xyz_gradient_accum.cpu().numpy(), z["stats_accum"], rtol=1e-6, atol=0)
    np.testing.assert_array_equal(d.denom.cpu().numpy(), z["stats_denom"])
    np.testing.assert_array_equal(d.max_radii2D.cpu().numpy(), z["stats_max_radii"])
    max_grad, min_opacity, extent, max_screen = (float(v) for v in z["params"][:4])
    d.densify_and_prune(max_grad, min_opacity, extent, max_screen)
    assert m._xyz.shape[0] == z["after_xyz"].shape[0]
    # everything except the sampled child positions is RNG-free
    for n in ("f_dc", "f_rest", "opacity", "scaling", "rotation"):
        np.testing.assert_allclose(getattr(m, NAMES[n]).detach().cpu().numpy(), z["after_" + n], rtol=0, atol=2e-6, err_msg=n)
