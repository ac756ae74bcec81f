#!/usr/bin/env python
"""Generate the committed golden fixtures by IMPORTING the reference's own Python
on CPU.  Run only in the build container:

    python tests/golden/make_goldens.py [/root/reference]

Nothing from the reference is copied: only seeded inputs and the numeric
outputs the reference code produced for them are written (``*.npz``).
The reference's CUDA rasterizer cannot run here, so ``render()`` is exercised
around a *fake* rasterizer returning seeded tensors — this pins the reference's
post-processing of ``allmap`` (gaussian_renderer/__init__.py:118-169), not the
kernels.
"""
import math
import os
import sys
import types

import numpy as np
import torch

_argv = [a for a in sys.argv[1:] if not a.startswith("--only=")]
ONLY = set(sum((a[len("--only="):].split(",") for a in sys.argv[1:] if a.startswith("--only=")), []))   # e.g. --only=transmat.npz
REF = _argv[0] if _argv else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
if not os.path.isdir(REF):
    print("reference not present; nothing to do")
    sys.exit(0)
sys.path.insert(0, REF)

# ---- stub third-party modules that are absent in this image -----------------
from typing import NamedTuple


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


for name in ["open3d", "cv2", "trimesh", "einsum", "lpips", "pyrender", "e3nn", "kornia", "plyfile"]:
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            _stub(name)
sys.modules["plyfile"].__dict__.setdefault("PlyData", object)
sys.modules["plyfile"].__dict__.setdefault("PlyElement", object)
sys.modules["kornia"].__dict__.setdefault("create_meshgrid", None)
sys.modules["e3nn"].__dict__.setdefault("o3", None)
_stub("simple_knn")
_stub("simple_knn._C", distCUDA2=None)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


FAKE = {}


class FakeRasterizer(torch.nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings
        FAKE["settings"] = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, extra_attrs=None):
        FAKE["extra_in"] = None if extra_attrs is None else extra_attrs.detach().clone()
        FAKE["cov3D_in"] = None if cov3D_precomp is None else cov3D_precomp.detach().clone()
        FAKE["cov3D_graph"] = cov3D_precomp         # still attached to the model's parameters
        FAKE["scales_in"], FAKE["rotations_in"] = scales, rotations
        return FAKE["color"], FAKE["radii"], FAKE["allmap"], FAKE["extra"], FAKE["grp"]


_stub("diff_surfel_rasterization", GaussianRasterizationSettings=GaussianRasterizationSettings,
      GaussianRasterizer=FakeRasterizer)

torch.Tensor.cuda = lambda self, *a, **k: self
_orig_tensor_to = torch.Tensor.to

from torch.overrides import TorchFunctionMode


class CpuMode(TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = kwargs or {}
        if "device" in kwargs and kwargs["device"] is not None and "cuda" in str(kwargs["device"]):
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


def save(name, **arrs):
    if ONLY and name not in ONLY:
        print("skipped", name)
        return
    arrs = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print("wrote", name, {k: v.shape for k, v in arrs.items()})


with CpuMode():
    # ------------------------------------------------------------------ G3
    from utils.contrastive_utils import contrastive_loss

    g = torch.Generator().manual_seed(1234)
    cases = {}
    for tag, (Nb, F, K, consider_negative, predef) in {
        "computed": (512, 16, 9, False, False),
        "predef": (384, 16, 12, False, True),
        "negative": (256, 8, 6, True, False),
        "f32dim": (1024, 32, 20, False, True),
    }.items():
        feats = torch.randn(Nb, F, generator=g)
        labels = torch.randint(0, K + 1, (Nb,), generator=g)      # 0 = unlabeled
        predef_u = None
        if predef:
            predef_u = torch.nn.functional.normalize(torch.randn(K + 1, F, generator=g), dim=1)
        f = feats.clone().requires_grad_(True)
        loss = contrastive_loss(f, labels, predef_u_list=predef_u, consider_negative=consider_negative)
        loss.backward()
        cases[f"{tag}_features"] = feats
        cases[f"{tag}_labels"] = labels
        cases[f"{tag}_predef"] = predef_u if predef_u is not None else torch.zeros(0)
        cases[f"{tag}_consider_negative"] = np.array(consider_negative)
        cases[f"{tag}_loss"] = loss.detach()
        cases[f"{tag}_grad"] = f.grad
    # min_pixnum variant
    feats = torch.randn(300, 8, generator=g)
    labels = torch.randint(0, 15, (300,), generator=g)
    f = feats.clone().requires_grad_(True)
    loss = contrastive_loss(f, labels, min_pixnum=18)
    loss.backward()
    cases.update(minpix_features=feats, minpix_labels=labels, minpix_loss=loss.detach(), minpix_grad=f.grad,
                 minpix_min_pixnum=np.array(18))
    save("contrastive_loss.npz", **cases)

    # ------------------------------------------------------------------ G3b: the reference's DEFAULT batch (arguments/__init__.py:65,103:
    # seg_feat_dim = 16, sample_batchsize = 32 768): rows of a pool drawn WITH replacement (train_semantic.py:183-190), the loss on the
    # drawn rows, and its gradient w.r.t. the POOL through the reference's own indexing (repeats accumulate).  The inputs are 2.6 MB and
    # regenerable from the seed (torch's CPU generator), so the fixture holds the seed, checksums of the inputs, the loss, and a digest
    # of the two gradients: 1 024 rows each, the column sums and the L1 norm.
    big = {}
    for tag, (seed, Nb, F, K, pool_n, predef) in {"computed": (4321, 32768, 16, 64, 20000, False),
                                                  "predef": (4322, 32768, 16, 64, 20000, True)}.items():
        gg = torch.Generator().manual_seed(seed)
        pool = torch.randn(pool_n, F, generator=gg)
        pool_labels = torch.randint(0, K + 1, (pool_n,), generator=gg)              # 0 = unlabeled
        idx = torch.randint(0, pool_n, (Nb,), generator=gg)
        predef_u = torch.nn.functional.normalize(torch.randn(K + 1, F, generator=gg), dim=1) if predef else None
        pick = torch.randperm(Nb, generator=gg)[:1024]
        pick_pool = torch.randperm(pool_n, generator=gg)[:1024]
        p_ = pool.clone().requires_grad_(True)
        f = p_[idx]
        f.retain_grad()
        loss = contrastive_loss(f, pool_labels[idx], predef_u_list=predef_u)
        loss.backward()
        big.update({f"{tag}_seed": np.array(seed), f"{tag}_dims": np.array([Nb, F, K, pool_n, int(predef)]),
                    f"{tag}_check": np.array([float(pool.double().sum()), float(pool.double().abs().sum()), float(idx.sum()), float(pool_labels.sum())]),
                    f"{tag}_loss": loss.detach(), f"{tag}_pick": pick, f"{tag}_pick_pool": pick_pool,
                    f"{tag}_grad_f_rows": f.grad[pick], f"{tag}_grad_f_colsum": f.grad.double().sum(0), f"{tag}_grad_f_l1": f.grad.double().abs().sum(),
                    f"{tag}_grad_pool_rows": p_.grad[pick_pool], f"{tag}_grad_pool_colsum": p_.grad.double().sum(0),
                    f"{tag}_grad_pool_l1": p_.grad.double().abs().sum(), f"{tag}_grad_pool_max": p_.grad.abs().max()})
    save("contrastive_loss_big.npz", **big)

    # ------------------------------------------------------------------ G5 cameras
    from scene.cameras import Camera
    from utils.graphics_utils import getProjectionMatrix, getWorld2View2

    cam_out = {}
    rng = np.random.RandomState(7)
    cams = []
    for i in range(4):
        A = rng.randn(3, 3)
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] = -Q[:, 0]
        T = rng.randn(3) * 2.0
        fovx = math.radians(40 + 15 * i)
        W, H = [(64, 48), (80, 48), (128, 96), (40, 40)][i]
        fovy = 2 * math.atan(math.tan(fovx / 2) * H / W)
        cam = Camera(colmap_id=i, R=Q, T=T, FoVx=fovx, FoVy=fovy, image=torch.zeros(3, H, W), image_name=str(i), uid=i,
                     data_device="cpu")
        cams.append(cam)
        cam_out.update({f"R{i}": Q, f"T{i}": T, f"fov{i}": np.array([fovx, fovy]), f"wh{i}": np.array([W, H]),
                        f"wvt{i}": cam.world_view_transform, f"proj{i}": cam.projection_matrix,
                        f"full{i}": cam.full_proj_transform, f"center{i}": cam.camera_center})
    save("cameras.npz", **cam_out)

    # ------------------------------------------------------------------ G4 render() post-processing
    import gaussian_renderer

    class FakePC:
        active_sh_degree = 3
        max_sh_degree = 3

        def __init__(self, P, F, gen):
            self._xyz = torch.randn(P, 3, generator=gen)
            self._seg = torch.randn(P, F, generator=gen) if F else None
            self._op = torch.rand(P, 1, generator=gen)
            self._sc = torch.rand(P, 2, generator=gen) * 0.1
            self._rot = torch.nn.functional.normalize(torch.randn(P, 4, generator=gen))
            self._sh = torch.randn(P, 16, 3, generator=gen)

        get_xyz = property(lambda s: s._xyz)
        get_opacity = property(lambda s: s._op)
        get_seg_feature = property(lambda s: s._seg)
        get_scaling = property(lambda s: s._sc)
        get_rotation = property(lambda s: s._rot)
        get_features = property(lambda s: s._sh)

    class Pipe:
        compute_cov3D_python = False
        convert_SHs_python = False
        depth_ratio = 1.0
        debug = False

    rp = {}
    for i, cam in enumerate(cams[:3]):
        gen = torch.Generator().manual_seed(100 + i)
        W, H = cam.image_width, cam.image_height
        P, F = 50, 6
        pc = FakePC(P, F, gen)
        alpha = torch.rand(1, H, W, generator=gen)
        alpha[:, : H // 4] = 0.0            # empty region -> nan_to_num path
        allmap = torch.cat([alpha * (1.0 + 3.0 * torch.rand(1, H, W, generator=gen)), alpha,
                            torch.randn(3, H, W, generator=gen) * alpha,
                            (1.0 + 3.0 * torch.rand(1, H, W, generator=gen)) * (alpha > 0),
                            torch.rand(1, H, W, generator=gen) * 0.01], dim=0)
        FAKE.update(color=torch.rand(3, H, W, generator=gen), radii=torch.randint(0, 5, (P,), generator=gen).int(),
                    allmap=allmap, extra=torch.randn(F, H, W, generator=gen),
                    grp=torch.randint(0, P, (17, 2), generator=gen).int())
        for ratio in (1.0, 0.0):
            pipe = Pipe()
            pipe.depth_ratio = ratio
            out = gaussian_renderer.render(cam, pc, pipe, torch.zeros(3))
            tag = f"c{i}_r{int(ratio)}"
            rp[f"{tag}_allmap"] = allmap
            for k in ["rend_alpha", "rend_normal", "rend_dist", "surf_depth", "surf_normal", "rend_depth",
                      "rend_median_depth", "visibility_filter"]:
                rp[f"{tag}_{k}"] = out[k]
        rp[f"c{i}_seg_raw"] = pc._seg
        rp[f"c{i}_seg_passed_to_rasterizer"] = FAKE["extra_in"]
        s = FAKE["settings"]
        rp[f"c{i}_settings"] = np.array([s.image_height, s.image_width, s.tanfovx, s.tanfovy, s.scale_modifier,
                                         s.sh_degree], dtype=np.float64)
    save("render_post.npz", **rp)

    # ------------------------------------------------------------------ K1 homography by the reference's own Python
    # pipe.compute_cov3D_python = True: render() builds transMat_precomp = cov3D_precomp itself
    # (gaussian_renderer/__init__.py:69-82) from GaussianModel.get_covariance (scene/gaussian_model.py:35-42,137-138).
    from scene.gaussian_model import GaussianModel as _RefGM

    tm = {}
    gen = torch.Generator().manual_seed(31337)
    Ptm = 300
    gm = _RefGM(3)
    gm.active_sh_degree = 3
    gm._xyz = torch.randn(Ptm, 3, generator=gen) * 1.2
    gm._scaling = torch.log(0.01 + 0.3 * torch.rand(Ptm, 2, generator=gen))
    gm._rotation = torch.randn(Ptm, 4, generator=gen) * (0.2 + 2.0 * torch.rand(Ptm, 1, generator=gen))   # NOT unit length
    gm._opacity = torch.randn(Ptm, 1, generator=gen)
    gm._features_dc = torch.randn(Ptm, 1, 3, generator=gen)
    gm._features_rest = torch.randn(Ptm, 15, 3, generator=gen) * 0.1
    gm._seg_feature = None
    if not hasattr(gm, "covariance_activation"):
        gm.setup_functions()
    tm.update(xyz=gm._xyz, log_scaling=gm._scaling, rotation_raw=gm._rotation)
    for i, cam in enumerate(cams[:4]):
        H_, W_ = cam.image_height, cam.image_width
        FAKE.update(color=torch.zeros(3, H_, W_), radii=torch.zeros(Ptm).int(), allmap=torch.ones(7, H_, W_),
                    extra=torch.zeros(0), grp=torch.zeros(0, 2).int())
        for mod in (1.0, 0.6, 1.7):
            pipe = Pipe()
            pipe.compute_cov3D_python = True
            try:
                gaussian_renderer.render(cam, gm, pipe, torch.zeros(3), scaling_modifier=mod)
            except Exception as e:      # the post-processing after the rasterizer call is not what is captured here
                print("render() after the rasterizer call:", repr(e))
            assert FAKE["cov3D_in"] is not None and FAKE["scales_in"] is None and FAKE["rotations_in"] is None
            tm[f"cam{i}_mod{mod:g}"] = FAKE["cov3D_in"]
    save("transmat.npz", **tm)

    # ------------------------------------------------------------------ K10 by the reference's own autograd
    # (a) dL/dtransMat -> dL/d(xyz, log-scaling, raw rotation): autograd through render()'s compute_cov3D_python graph
    #     (gaussian_renderer/__init__.py:69-82, scene/gaussian_model.py:35-42, utils/general_utils.py build_rotation);
    # (b) dL/dcolor -> dL/d(SH, xyz): autograd through the reference's own eval_sh (utils/sh_utils.py:57-117) composed as the
    #     convert_SHs_python branch of render() composes it (:92-97: direction, eval_sh, + 0.5, clamp_min 0).
    from utils.sh_utils import eval_sh as _eval_sh
    kb = {}
    gen = torch.Generator().manual_seed(424242)
    for i, cam in enumerate(cams[:4]):
        for prm in (gm._xyz, gm._scaling, gm._rotation):
            prm.requires_grad_(True)
            prm.grad = None
        H_, W_ = cam.image_height, cam.image_width
        FAKE.update(color=torch.zeros(3, H_, W_), radii=torch.zeros(Ptm).int(), allmap=torch.ones(7, H_, W_),
                    extra=torch.zeros(0), grp=torch.zeros(0, 2).int())
        pipe = Pipe()
        pipe.compute_cov3D_python = True
        try:
            gaussian_renderer.render(cam, gm, pipe, torch.zeros(3), scaling_modifier=1.0)
        except Exception as e:
            print("render() after the rasterizer call:", repr(e))
        dT = torch.randn(Ptm, 9, generator=gen)
        (FAKE["cov3D_graph"] * dT).sum().backward()
        kb[f"cam{i}_dL_dtransMat"] = dT
        kb[f"cam{i}_grad_xyz"] = gm._xyz.grad.clone()
        kb[f"cam{i}_grad_log_scaling"] = gm._scaling.grad.clone()
        kb[f"cam{i}_grad_rotation_raw"] = gm._rotation.grad.clone()
        # (b)
        shs = torch.cat((gm._features_dc, gm._features_rest), dim=1).detach().clone().requires_grad_(True)      # [P,16,3]
        xyz2 = gm._xyz.detach().clone().requires_grad_(True)
        for deg in (1, 3):
            shs.grad = None; xyz2.grad = None
            shs_view = shs.transpose(1, 2).view(-1, 3, 16)
            dir_pp = xyz2 - cam.camera_center.repeat(Ptm, 1)
            dirn = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            col = torch.clamp_min(_eval_sh(deg, shs_view, dirn) + 0.5, 0.0)
            dC_ = torch.randn(Ptm, 3, generator=gen)
            (col * dC_).sum().backward()
            kb[f"cam{i}_deg{deg}_dL_dcolor"] = dC_
            kb[f"cam{i}_deg{deg}_color"] = col.detach().clone()
            kb[f"cam{i}_deg{deg}_grad_shs"] = shs.grad.clone()
            kb[f"cam{i}_deg{deg}_grad_xyz"] = xyz2.grad.clone()
    kb["shs"] = torch.cat((gm._features_dc, gm._features_rest), dim=1).detach()
    for prm in (gm._xyz, gm._scaling, gm._rotation):
        prm.requires_grad_(False)
    save("kten_backward.npz", **kb)

    # ------------------------------------------------------------------ SH + rotation helpers
    from utils.sh_utils import eval_sh
    from utils.general_utils import build_rotation

    gen = torch.Generator().manual_seed(55)
    shs = torch.randn(200, 16, 3, generator=gen)
    dirs = torch.nn.functional.normalize(torch.randn(200, 3, generator=gen), dim=1)
    sh_out = {"shs": shs, "dirs": dirs}
    for deg in range(4):
        sh_out[f"rgb_deg{deg}"] = eval_sh(deg, shs.transpose(1, 2), dirs)
    q = torch.randn(200, 4, generator=gen)
    sh_out["quats"] = q
    sh_out["rotmats"] = build_rotation(q)
    save("sh_rot.npz", **sh_out)

    # ------------------------------------------------------------------ losses (H2)
    from utils.loss_utils import l1_loss, ssim

    gen = torch.Generator().manual_seed(77)
    a = torch.rand(3, 40, 56, generator=gen).requires_grad_(True)
    b = torch.rand(3, 40, 56, generator=gen)
    l1 = l1_loss(a, b)
    ss = ssim(a, b)
    (0.8 * l1 + 0.2 * (1.0 - ss)).backward()
    save("losses.npz", img=a.detach(), gt=b, l1=l1.detach(), ssim=ss.detach(), grad=a.grad)

    # ------------------------------------------------------------------ depth_to_normal
    from utils.point_utils import depth_to_normal

    dn = {}
    for i, cam in enumerate(cams[:2]):
        gen = torch.Generator().manual_seed(200 + i)
        depth = 1.0 + 2.0 * torch.rand(1, cam.image_height, cam.image_width, generator=gen)
        dn[f"depth{i}"] = depth
        dn[f"normal{i}"] = depth_to_normal(cam, depth)
    save("depth_to_normal.npz", **dn)

    # ------------------------------------------------------------------ Gram-Schmidt class features (L3)
    from scene.gaussian_model import GaussianModel

    try:
        gm = GaussianModel(3)
        gm.seg_feat_dim = 16
        gm._seg_feature = None
        gm._xyz = torch.zeros(40, 3)
        torch.manual_seed(9)
        masks = np.zeros((40, 5), dtype=bool)
        for k in range(5):
            masks[k * 8:(k + 1) * 8, k] = True
        state = torch.get_rng_state()
        gm.set_3d_feat(masks, gram_feat=True)
        torch.set_rng_state(state)
        seg0 = torch.rand((40, 16))
        init = torch.rand((5, 16))
        save("gram_schmidt.npz", init_rand=init, class_feat=gm.class_feat, seg_feature=gm._seg_feature.detach(),
             seg_rand=seg0, masks=masks)
    except Exception as e:  # pragma: no cover
        print("gram-schmidt golden skipped:", repr(e))

    # ------------------------------------------------------------------ densification (train.py:138-151)
    try:
        class _Args:
            percent_dense = 0.01
            position_lr_init = 0.00016
            position_lr_final = 0.0000016
            position_lr_delay_mult = 0.01
            position_lr_max_steps = 30_000
            feature_lr = 0.0025
            opacity_lr = 0.05
            scaling_lr = 0.005
            rotation_lr = 0.001
            seg_feature_lr = 0.025

        gen = torch.Generator().manual_seed(4242)
        P = 600
        gm = GaussianModel(3)
        gm.spatial_lr_scale = 1.0
        mk = lambda t: torch.nn.Parameter(t.requires_grad_(True))
        init = dict(xyz=torch.randn(P, 3, generator=gen), f_dc=torch.randn(P, 1, 3, generator=gen),
                    f_rest=0.1 * torch.randn(P, 15, 3, generator=gen), opacity=1.5 * torch.randn(P, 1, generator=gen),
                    scaling=math.log(0.03) + 0.8 * torch.randn(P, 2, generator=gen), rotation=torch.randn(P, 4, generator=gen))
        gm._xyz, gm._features_dc, gm._features_rest = mk(init["xyz"].clone()), mk(init["f_dc"].clone()), mk(init["f_rest"].clone())
        gm._opacity, gm._scaling, gm._rotation = mk(init["opacity"].clone()), mk(init["scaling"].clone()), mk(init["rotation"].clone())
        gm.max_radii2D = torch.zeros(P)
        gm.training_setup(_Args())
        out = {"init_" + k: v for k, v in init.items()}
        # two optimiser steps with seeded gradients (non-trivial Adam moments), three rounds of statistics
        names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
        params = [gm._xyz, gm._features_dc, gm._features_rest, gm._opacity, gm._scaling, gm._rotation]
        for s in range(2):
            for n, p_ in zip(names, params):
                p_.grad = 0.01 * torch.randn(p_.shape, generator=gen)
                out[f"grad{s}_{n}"] = p_.grad.clone()
            gm.optimizer.step()
            gm.optimizer.zero_grad(set_to_none=True)
        for s in range(3):
            vs = torch.zeros(P, 3, requires_grad=True)
            vs.grad = torch.cat([0.0008 * torch.rand(P, 2, generator=gen), torch.zeros(P, 1)], dim=1)
            vis = torch.rand(P, generator=gen) > 0.3
            radii = torch.randint(0, 40, (P,), generator=gen).int()
            gm.max_radii2D[vis] = torch.max(gm.max_radii2D[vis], radii[vis])
            gm.add_densification_stats(vs, vis)
            out[f"vsgrad{s}"], out[f"vis{s}"], out[f"radii{s}"] = vs.grad.clone(), vis, radii
        out["stats_accum"], out["stats_denom"], out["stats_max_radii"] = gm.xyz_gradient_accum.clone(), gm.denom.clone(), gm.max_radii2D.clone()
        torch.manual_seed(31337)
        gm.densify_and_prune(0.0002, 0.05, 2.5, 20)
        for n, attr in zip(names, ["_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"]):
            p_ = getattr(gm, attr)
            st = gm.optimizer.state[p_]
            out[f"after_{n}"], out[f"after_m_{n}"], out[f"after_v_{n}"] = p_.detach(), st["exp_avg"], st["exp_avg_sq"]
        out["after_accum"], out["after_denom"], out["after_max_radii"] = gm.xyz_gradient_accum, gm.denom, gm.max_radii2D
        gm.reset_opacity()
        st = gm.optimizer.state[gm._opacity]
        out["reset_opacity"], out["reset_m"], out["reset_v"] = gm._opacity.detach(), st["exp_avg"], st["exp_avg_sq"]
        out["params"] = np.array([0.0002, 0.05, 2.5, 20, 0.01])     # max_grad, min_opacity, extent, max_screen_size, percent_dense
        out["seed"] = np.array(31337)
        save("densify.npz", **out)
    except Exception as e:  # pragma: no cover
        import traceback
        traceback.print_exc()
        print("densification golden skipped:", repr(e))

    # ------------------------------------------------------------------ PLY checkpoints (scene/gaussian_model.py:263-321,364-422)
    # `plyfile` is not installed: its two entry points are replaced by RECORDING stand-ins for the duration of this block, so
    # that the reference's own save_ply / load_ply run and what they hand to / ask from plyfile is captured - the structured
    # array of the vertex element (property names in order, float32 values: everything a binary_little_endian PLY of float
    # properties consists of besides its header syntax) and, on the way back, the parameter tensors load_ply builds from such a
    # file.  The container syntax itself (header lines, little-endian packing) is the public PLY format, not the reference's.
    try:
        import scene.gaussian_model as _gmod

        class _Rec:
            elements = None

        class _El:
            def __init__(self, data, name):
                self.data, self.name = data, name
                self.properties = [types.SimpleNamespace(name=n) for n in data.dtype.names]

            def __getitem__(self, key):
                return self.data[key]

        class _PlyElement:
            @staticmethod
            def describe(data, name):
                return _El(np.array(data), name)

        class _PlyData:
            def __init__(self, elements):
                self.elements = list(elements)

            def write(self, path):
                _Rec.elements = self.elements[0].data

            @staticmethod
            def read(path):
                return _PlyData([_El(_Rec.elements, "vertex")])

        class _O3D:         # the two colour previews the reference also writes with open3d are not part of the checkpoint
            class geometry:
                class PointCloud:
                    points = colors = None

            class utility:
                Vector3dVector = staticmethod(lambda a: a)

            class io:
                write_point_cloud = staticmethod(lambda *a, **k: None)

        saved = (_gmod.PlyData, _gmod.PlyElement, _gmod.o3d, _gmod.mkdir_p, _gmod.feature3d_to_rgb)
        _gmod.PlyData, _gmod.PlyElement, _gmod.o3d, _gmod.mkdir_p = _PlyData, _PlyElement, _O3D, (lambda d: None)
        _gmod.feature3d_to_rgb = lambda f: np.zeros((f.shape[0], 3))
        try:
            gen = torch.Generator().manual_seed(777)
            out = {}
            for tag, Fd in (("feat", 6), ("nofeat", 0)):
                P = 23
                gm = _gmod.GaussianModel(3)
                init = dict(xyz=torch.randn(P, 3, generator=gen), f_dc=torch.randn(P, 1, 3, generator=gen),
                            f_rest=torch.randn(P, 15, 3, generator=gen), opacity=torch.randn(P, 1, generator=gen),
                            scaling=torch.randn(P, 2, generator=gen), rotation=torch.randn(P, 4, generator=gen))
                gm._xyz, gm._features_dc, gm._features_rest = init["xyz"], init["f_dc"], init["f_rest"]
                gm._opacity, gm._scaling, gm._rotation = init["opacity"], init["scaling"], init["rotation"]
                gm._seg_feature = torch.randn(P, Fd, generator=gen) if Fd else None
                if Fd:
                    init["seg_feature"] = gm._seg_feature
                crop = (torch.arange(P) % 4 != 1) if tag == "nofeat" else None
                gm.save_ply("golden/point_cloud.ply", crop_mask=crop)
                el = _Rec.elements
                out.update({f"{tag}_in_{k}": v for k, v in init.items()})
                if crop is not None:
                    out[f"{tag}_crop"] = crop
                out[f"{tag}_names"] = np.array(list(el.dtype.names))
                out[f"{tag}_body"] = np.frombuffer(el.astype([(n, "<f4") for n in el.dtype.names]).tobytes(), dtype=np.uint8)
                # and back through the reference's load_ply (it reads `use_seg_feature`, `load_seg_feat`, `seg_feat_dim`)
                gl = _gmod.GaussianModel(3)
                gl.use_seg_feature, gl.load_seg_feat, gl.seg_feat_dim = bool(Fd), bool(Fd), Fd
                gl.load_ply("golden/point_cloud.ply")
                for k, attr in (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
                                ("scaling", "_scaling"), ("rotation", "_rotation")):
                    out[f"{tag}_loaded_{k}"] = getattr(gl, attr).detach()
                if Fd:
                    out[f"{tag}_loaded_seg_feature"] = gl._seg_feature.detach()
            save("ply.npz", **out)
        finally:
            _gmod.PlyData, _gmod.PlyElement, _gmod.o3d, _gmod.mkdir_p, _gmod.feature3d_to_rgb = saved
    except Exception as e:  # pragma: no cover
        import traceback
        traceback.print_exc()
        print("ply golden skipped:", repr(e))

    # ------------------------------------------------------------------ tracer consumer (spatial_track/modules/init_tracker.py:16-47)
    try:
        import spatial_track.modules.init_tracker as _trk

        gen = torch.Generator().manual_seed(99)
        H, W, P, K = 40, 56, 3000, 20000
        seg = torch.randint(0, 7, (H, W), generator=gen)
        seg[seg == 6] = 5
        seg[0, :3] = 6                                 # a mask with fewer than 50 distinct Gaussians: dropped (:40-41)
        grp = torch.stack([torch.randint(0, P, (K,), generator=gen), torch.randint(0, H * W, (K,), generator=gen)], 1).int()
        _saved_render = _trk.render
        _trk.render = lambda view, gaussian, pipe, bg: {"gau_related_pixels": grp.long()}
        try:
            view = types.SimpleNamespace(segmap=seg)
            gaussian = types.SimpleNamespace(pipelineparams=None)
            info, frame_ids = _trk.get_segmap_gaussians(gaussian, view)
        finally:
            _trk.render = _saved_render
        keys = sorted(int(k) for k in info)
        out = {"segmap": seg, "gau_related_pixels": grp, "mask_ids": np.array(keys), "frame_ids": np.array(sorted(frame_ids))}
        for k in keys:
            out[f"mask_{k}"] = np.array(sorted(info[k]))
        save("tracker.npz", **out)
    except Exception as e:  # pragma: no cover
        import traceback
        traceback.print_exc()
        print("tracker golden skipped:", repr(e))

    # ------------------------------------------------------------------ optimisation defaults (arguments/__init__.py:89-127)
    try:
        from argparse import ArgumentParser
        from arguments import OptimizationParams
        op = OptimizationParams(ArgumentParser())
        keys = ["seg_feature_lr", "sample_mv_frames", "lambda_singview_contras", "lambda_multiview_contras", "lambda_3D_contras",
                "lambda_dssim", "lambda_dist", "lambda_normal", "percent_dense", "opacity_cull", "densification_interval",
                "opacity_reset_interval", "densify_from_iter", "densify_until_iter", "densify_grad_threshold", "position_lr_init",
                "feature_lr", "opacity_lr", "scaling_lr", "rotation_lr"]
        save("defaults.npz", **{k: np.array(float(getattr(op, k))) for k in keys})
    except Exception as e:  # pragma: no cover
        print("defaults golden skipped:", repr(e))

    # ------------------------------------------------------------------ COLMAP sparse model
    try:
        import tempfile
        sys.path.insert(0, os.path.join(os.path.dirname(OUT), ".."))
        from instascene_amd import colmap_io as cio          # only its WRITERS are used here, to make the input files
        from scene import colmap_loader as ref_cl
        from scene.dataset_readers import getNerfppNorm
        from utils.graphics_utils import focal2fov

        rng = np.random.RandomState(21)
        intr = {1: cio.Intrinsics(1, "PINHOLE", 640, 480, np.array([520.5, 515.25, 320.0, 240.0])),
                7: cio.Intrinsics(7, "SIMPLE_PINHOLE", 800, 600, np.array([700.0, 400.0, 300.0])),
                3: cio.Intrinsics(3, "OPENCV", 320, 200, np.array([250.0, 260.0, 160.0, 100.0, 0.01, -0.02, 0.0, 0.0]))}
        poses, obs = {}, {}
        for k, iid in enumerate([4, 9, 2, 11, 6]):
            q = rng.randn(4); q /= np.linalg.norm(q)
            poses[iid] = cio.Pose(iid, q, rng.randn(3) * 2.0, [1, 7, 3, 1, 7][k], f"sub/frame_{20 - k:03d}.jpg")
            obs[iid] = np.concatenate([rng.rand(k * 3, 2) * 100, rng.randint(-1, 50, (k * 3, 1))], axis=1)
        n = 40
        xyz, rgb, err = rng.randn(n, 3), rng.randint(0, 256, (n, 3)).astype(np.uint8), rng.rand(n)
        tracks = [np.stack([rng.randint(1, 12, t), rng.randint(0, 30, t)], axis=1) for t in rng.randint(0, 6, n)]
        with tempfile.TemporaryDirectory() as td:
            cio.write_cameras_bin(os.path.join(td, "cameras.bin"), intr)
            cio.write_images_bin(os.path.join(td, "images.bin"), poses, obs)
            cio.write_points3d_bin(os.path.join(td, "points3D.bin"), xyz, rgb, err, tracks)
            files = {k: np.frombuffer(open(os.path.join(td, k + ".bin"), "rb").read(), dtype=np.uint8) for k in ("cameras", "images", "points3D")}
            r_ext = ref_cl.read_extrinsics_binary(os.path.join(td, "images.bin"))
            r_int = ref_cl.read_intrinsics_binary(os.path.join(td, "cameras.bin"))
            r_xyz, r_rgb, r_err = ref_cl.read_points3D_binary(os.path.join(td, "points3D.bin"))
        out = {"file_" + k: v for k, v in files.items()}
        infos = []

        class _CI:
            pass

        for key in r_ext:
            e = r_ext[key]
            i_ = r_int[e.camera_id]
            ci = _CI()
            ci.R, ci.T = np.transpose(ref_cl.qvec2rotmat(e.qvec)), np.array(e.tvec)
            fy = i_.params[0] if i_.model in ("SIMPLE_PINHOLE", "SIMPLE_RADIAL") else i_.params[1]
            ci.FovY, ci.FovX = focal2fov(fy, i_.height), focal2fov(i_.params[0], i_.width)
            ci.name, ci.uid, ci.width, ci.height = os.path.basename(e.name).split(".")[0], i_.id, i_.width, i_.height
            infos.append(ci)
        infos.sort(key=lambda c: c.name)
        norm = getNerfppNorm(infos)
        out.update(names=np.array([c.name for c in infos]), uid=np.array([c.uid for c in infos]),
                   R=np.stack([c.R for c in infos]), T=np.stack([c.T for c in infos]),
                   fov=np.array([[c.FovX, c.FovY] for c in infos]), wh=np.array([[c.width, c.height] for c in infos]),
                   xyz=r_xyz, rgb=r_rgb, err=r_err, norm_translate=norm["translate"], norm_radius=np.array(norm["radius"]))
        mats = []
        for j, c in enumerate(infos):
            cam = Camera(colmap_id=c.uid, R=c.R, T=c.T, FoVx=c.FovX, FoVy=c.FovY, image=torch.zeros(3, c.height, c.width),
                         image_name=c.name, uid=j, data_device="cpu")
            mats.append(torch.stack([cam.world_view_transform, cam.full_proj_transform]))
            out[f"center{j}"] = cam.camera_center
        out["matrices"] = torch.stack(mats)
        save("colmap.npz", **out)
    except Exception as e:  # pragma: no cover
        import traceback
        traceback.print_exc()
        print("colmap golden skipped:", repr(e))
