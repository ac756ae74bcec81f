"""``render()`` — mirror of the reference's ``gaussian_renderer.render``
(gaussian_renderer/__init__.py:20-169): same signature, same returned dict
(13 entries), built on the HIP rasterizer.  The post-processing of the 7-channel
``allmap`` (:127-167) incl. ``depth_to_normal`` (utils/point_utils.py:10-40) is
``csrc/iso_post.hip`` (two kernels each way)."""
from __future__ import annotations

import ctypes
import math
import os

import torch

from . import _hot, _lib
from .contrastive import row_normalize
from . import rasterizer as _rz
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


_ENV_LAZY_MAPS = os.environ.get("ISR_LAZY_MAPS", "0") == "1"
_LAZY_KEYS = ("rend_alpha", "rend_normal", "rend_dist", "surf_depth", "surf_normal", "rend_depth", "rend_median_depth")


class RenderPackage(dict):
    """The dict ``render()`` returns (same 13 keys as the reference).

    * ``gau_related_pixels`` is sliced to its valid length on first access (the slice needs the count on the host,
      i.e. a device sync — the reference pays it inside every forward, :106).
    * Opt-in (``pipe.lazy_maps = True`` or ``ISR_LAZY_MAPS=1``): the seven maps derived from ``allmap`` (:127-167)
      are evaluated on first access of any of them, under the grad mode ``render()`` was called in
      (``train_semantic`` never reads them).  Default: inside ``render()`` like the reference."""

    _pending = None

    def _materialize(self):
        job = self._pending
        if job is not None:
            self._pending = None
            cam, allmap, depth_ratio, grad_mode = job
            with torch.set_grad_enabled(grad_mode):
                dict.update(self, post_process(cam, allmap, depth_ratio))

    def __getitem__(self, k):
        if self._pending is not None and k in _LAZY_KEYS:
            self._materialize()
        v = dict.__getitem__(self, k)
        if v is None and k == "visibility_filter":
            # `radii > 0` (reference :110) on first access: the same values, one elementwise kernel over P Gaussians that a
            # loop which never reads it (a warmed-up SegTrainer) does not wait for between the forward and its losses
            v = dict.__getitem__(self, "radii") > 0
            dict.__setitem__(self, k, v)
        if k == "gau_related_pixels" and getattr(v, "_isr_last_index", None) is not None:
            v = _rz.slice_tracer(v)
            dict.__setitem__(self, k, v)
        return v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def __iter__(self):          # also routes dict(pkg) / {**pkg} through __getitem__
        self._materialize()
        if dict.get(self, "visibility_filter", 0) is None:
            self["visibility_filter"]
        return dict.__iter__(self)

    def items(self):
        return [(k, self[k]) for k in dict.keys(self)]

    def values(self):
        return [self[k] for k in dict.keys(self)]


_RAY_CACHE = {}
_ZEROS = {}


def _camera_rays(view, device):
    """The camera's per-pixel ray table for ``iso_render_post_forward`` (what utils/point_utils.py:10-27 rebuilds on
    every call; static per camera, so cached): ``rays_d[y*W + x]`` = world-space direction of the ray through pixel
    ``(x, y)`` scaled to unit view depth, ``rays_o`` = camera centre, i.e. a surface point is ``depth * rays_d + rays_o``.

    With the reference's row-vector storage (``p_view = p_world @ V``, ``p_clip = p_world @ VP``) the camera->clip map is
    ``Pj = V^-1 VP`` and a camera-space point lands on the homogeneous pixel ``q = p_cam @ K`` with
    ``K[:, 0] = W/2 (Pj[:3, 0] + Pj[:3, 3])``, ``K[:, 1] = H/2 (Pj[:3, 1] + Pj[:3, 3])``, ``K[:, 2] = Pj[:3, 3]``
    (the reference's pixel mapping for this table has offset W/2, not (W-1)/2).  Hence
    ``rays_d(x, y) = (x, y, 1) @ M`` with the single 3x3 matrix ``M = K^-1 (V^-1)[:3, :3]``: an affine function of the
    pixel, evaluated by broadcasting instead of a meshgrid and two [N,3]x[3,3] products."""
    key = (id(view), str(device))
    hit = _RAY_CACHE.get(key)
    if hit is not None and hit[0] is view.world_view_transform:
        return hit[1], hit[2]
    W, H = int(view.image_width), int(view.image_height)
    V = view.world_view_transform.to(device=device, dtype=torch.float64)
    VP = view.full_proj_transform.to(device=device, dtype=torch.float64)
    Vinv = torch.linalg.inv(V)
    Pj = (Vinv @ VP)[:3]
    K = torch.stack((0.5 * W * (Pj[:, 0] + Pj[:, 3]), 0.5 * H * (Pj[:, 1] + Pj[:, 3]), Pj[:, 3]), dim=1)
    M = (torch.linalg.inv(K) @ Vinv[:3, :3]).to(torch.float32)
    xs = torch.arange(W, device=device, dtype=torch.float32).view(1, W, 1)
    ys = torch.arange(H, device=device, dtype=torch.float32).view(H, 1, 1)
    rays_d = (xs * M[0] + ys * M[1] + M[2]).reshape(-1, 3).contiguous()
    rays_o = Vinv[3, :3].to(torch.float32).contiguous()
    if len(_RAY_CACHE) > 256:
        _RAY_CACHE.clear()
    _RAY_CACHE[key] = (view.world_view_transform, rays_d, rays_o)
    return rays_d, rays_o


class _RenderPost(torch.autograd.Function):
    """The seven derived maps in two HIP kernels each way (``iso_render_post_forward/backward``)."""

    @staticmethod
    def forward(ctx, allmap, viewmatrix, rays_d, rays_o, depth_ratio):
        L = _lib.lib()
        am = allmap.contiguous().float()
        _, H, W = am.shape
        # one allocation, seven [C,H,W] tensors on it - tensors of their own on the block's storage (Tensor.set_), NOT views of
        # it: autograd forbids in-place operations on views a custom Function returns, and user code may clamp / scale the
        # maps in place as it can with the reference's (gaussian_renderer/__init__.py:127-167 builds them with torch ops)
        block = torch.empty((11, H, W), dtype=torch.float32, device=am.device)
        st, parts, ch = block.untyped_storage(), [], 0
        for c in (1, 3, 1, 1, 3, 1, 1):
            parts.append(torch.empty(0, dtype=torch.float32, device=am.device).set_(st, block.storage_offset() + ch * H * W, (c, H, W)))
            ch += c
        alpha, normal, dist, surf, snorm, depth, median = parts
        vm = viewmatrix.contiguous().float()
        with _hot.on_device(am.device):
            _lib.check(L.iso_render_post_forward(W, H, float(depth_ratio), _ptr(am), _ptr(vm), _ptr(rays_d), _ptr(rays_o),
                                                 _ptr(alpha), _ptr(normal), _ptr(dist), _ptr(surf), _ptr(snorm),
                                                 _ptr(depth), _ptr(median), _stream()), "iso_render_post_forward")
        ctx.save_for_backward(am, vm, rays_d, rays_o, surf)
        ctx.ratio = float(depth_ratio)
        ctx.set_materialize_grads(False)
        return alpha, normal, dist, surf, snorm, depth, median

    @staticmethod
    def backward(ctx, g_alpha, g_normal, g_dist, g_surf, g_snorm, g_depth, g_median):
        am, vm, rays_d, rays_o, surf = ctx.saved_tensors
        if all(g is None for g in (g_alpha, g_normal, g_dist, g_surf, g_snorm, g_depth, g_median)):
            return None, None, None, None, None
        L = _lib.lib()
        _, H, W = am.shape
        c = lambda g: None if g is None else g.contiguous().float()
        g_alpha, g_normal, g_dist, g_surf = c(g_alpha), c(g_normal), c(g_dist), c(g_surf)
        g_snorm, g_depth, g_median = c(g_snorm), c(g_depth), c(g_median)
        scratch = torch.empty((6, H, W), dtype=torch.float32, device=am.device) if g_snorm is not None else None
        out = torch.empty_like(am)
        with _hot.on_device(am.device):
            _lib.check(L.iso_render_post_backward(W, H, ctx.ratio, _ptr(am), _ptr(vm), _ptr(rays_d), _ptr(rays_o), _ptr(surf),
                                                  _ptr(g_alpha), _ptr(g_normal), _ptr(g_dist), _ptr(g_surf), _ptr(g_snorm),
                                                  _ptr(g_depth), _ptr(g_median), _ptr(scratch), _ptr(out), _stream()),
                       "iso_render_post_backward")
        return out, None, None, None, None


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return _hot.stream_ptr()


def post_process(viewpoint_camera, allmap, depth_ratio):
    """The seven maps render() derives from the rasterizer's 7-channel ``allmap`` (gaussian_renderer/__init__.py:127-167,
    utils/point_utils.py:10-40): two HIP kernels each way.  Device tensors only - there is no host path in the product
    (the CPU restatement that tests/golden/render_post.npz pins lives in oracle/torch_ops.py)."""
    if not allmap.is_cuda:
        raise RuntimeError("instascene_amd.render.post_process: allmap must be a CUDA tensor (no CPU fallback)")
    rays_d, rays_o = _camera_rays(viewpoint_camera, allmap.device)
    alpha, normal, dist, surf, snorm, depth, median = _RenderPost.apply(
        allmap, viewpoint_camera.world_view_transform, rays_d, rays_o, float(depth_ratio))
    return {'rend_alpha': alpha, 'rend_normal': normal, 'rend_dist': dist, 'surf_depth': surf,
            'surf_normal': snorm, 'rend_depth': depth, 'rend_median_depth': median}


def _precomputed_transforms(viewpoint_camera, pc, scaling_modifier):
    """``pipe.compute_cov3D_python`` (gaussian_renderer/__init__.py:69-82): the per-Gaussian 3x3 splat->pixel homography
    built in torch from the model's 4x4 splat->world matrices (``pc.get_covariance``, scene/gaussian_model.py:35-42) and
    handed to the rasterizer as ``cov3D_precomp [P,9]`` (rows Tu, Tv, Tw).  Only rows/columns (0, 1, 3) of the 4x4
    factors take part - the splat has no third axis and the pixel has no depth row - so the NDC->pixel map is built as
    the [4,3] matrix it is used as."""
    W, H = float(viewpoint_camera.image_width), float(viewpoint_camera.image_height)
    vp = viewpoint_camera.full_proj_transform
    to_pixel = vp.new_tensor([[0.5 * W, 0.0, 0.0], [0.0, 0.5 * H, 0.0], [0.0, 0.0, 0.0], [0.5 * (W - 1.0), 0.5 * (H - 1.0), 1.0]])
    world_to_pixel = vp @ to_pixel                                            # [4,3]
    splat_rows = pc.get_covariance(scaling_modifier)[:, (0, 1, 3), :]         # [P,3,4]: tangent u, tangent v, centre
    return torch.einsum("prk,kc->pcr", splat_rows, world_to_pixel).reshape(-1, 9)


def _geometry_inputs(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color):
    """The rasterizer module of a view and the geometry/appearance arguments render() passes to it (:35-100)."""
    xyz = pc.get_xyz
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)
    scales = rotations = cov3D_precomp = None
    if getattr(pipe, "compute_cov3D_python", False):
        cov3D_precomp = _precomputed_transforms(viewpoint_camera, pc, scaling_modifier)
    else:
        scales = pc.get_scaling
        rotations = pc.get_rotation
    shs = colors_precomp = None
    if override_color is None:
        shs = pc.get_features          # the reference forces convert_SHs_python = False (:88)
    else:
        colors_precomp = override_color
    return GaussianRasterizer(raster_settings=raster_settings), dict(
        means3D=xyz, shs=shs, colors_precomp=colors_precomp, opacities=pc.get_opacity, scales=scales, rotations=rotations,
        cov3D_precomp=cov3D_precomp)


def prefetch(viewpoint_camera, pc, pipe, bg_color=None, scaling_modifier=1.0, override_color=None, stream=None,
             after=None) -> bool:
    """Issue the geometry pass and binning (projection, tile counts, scan, key scatter, tile sort) of the NEXT
    ``render()`` of this view now (extension).  They depend on the Gaussians' geometry and SH only, so a data-parallel
    trainer runs them next to the rest of the current step (``stream`` / ``after``: see
    ``rasterizer.prefetch_geometry``); ``render()`` then starts at the blend kernel.  No effect (False) without async
    binning."""
    rasterizer, geo = _geometry_inputs(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color)
    return rasterizer.prefetch(**geo, stream=stream, after=after)


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None,
           norm_seg_feat=True, sample_pixels=None, defer_rows=True):
    """Render the scene (reference gaussian_renderer/__init__.py:20).  ``pc`` needs the reference
    ``GaussianModel`` getters (get_xyz, get_opacity, get_scaling, get_rotation, get_features,
    get_seg_feature, active_sh_degree); background tensor must be on the GPU.  ``sample_pixels`` (extension, int64
    ``y*W + x``): also return ``sampled_seg_feature [n, F]``, the feature map at those pixels — what train_semantic.py
    builds by indexing the map (:118-129) — whose gradient reaches the Gaussians without a dense ``[F,H,W]`` map."""
    xyz = pc.get_xyz
    # The reference always makes this carrier require grad (:29-33).  Its gradient is only consumed by
    # train.py's densification; when the geometry is frozen (train_semantic) nothing reads it, so the
    # carrier follows xyz and the rasterizer can run its feature-only backward.
    need_geom_grad = bool(xyz.requires_grad) or bool(getattr(pipe, "force_viewspace_grad", False))
    # a constant: one zero tensor per (shape, device) instead of a fill per call (nothing reads or writes its values; with
    # geometry gradients each call gets its own LEAF over that storage, whose .grad the rasterizer's backward fills)
    zkey = (tuple(xyz.shape), xyz.dtype, str(xyz.device))
    screenspace_points = _ZEROS.get(zkey)
    if screenspace_points is None:
        if len(_ZEROS) > 8:
            _ZEROS.clear()
        screenspace_points = _ZEROS[zkey] = torch.zeros_like(xyz, requires_grad=False)
    if need_geom_grad:
        screenspace_points = screenspace_points.detach().requires_grad_(True)
    rasterizer, geo = _geometry_inputs(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color)

    means2D = screenspace_points
    seg_feature = pc.get_seg_feature
    if seg_feature is not None and norm_seg_feat:
        seg_feature = row_normalize(seg_feature, 1e-9)      # reference :61-62

    # opt-in (pipe.feature_only_forward, like lazy_maps): a trainer that reads nothing but the feature map - the whole of
    # train_semantic.py's loss - lets the blend kernel skip colour, the seven auxiliary maps and the tracer.  The dict keeps
    # all 13 keys; the skipped ones are None.  Never the default: the reference renders everything on every call.
    feature_only = (bool(getattr(pipe, "feature_only_forward", False)) and seg_feature is not None and not need_geom_grad
                    and _rz._CONFIG["mode"] == _lib.MODE_FAST and seg_feature.shape[1] > 0)
    # lazy_tracer: the tracer list is sliced to its valid length (a host sync) on first access of the dict entry
    res = rasterizer(means2D=means2D, extra_attrs=seg_feature, sample_pixels=sample_pixels if seg_feature is not None else None,
                     lazy_tracer=True, feature_only=feature_only, defer_rows=defer_rows, **geo)
    rendered_image, radii, allmap, extra_attrs, gau_related_pixels = res[:5]
    if feature_only:
        rets = RenderPackage({"render": None, "viewspace_points": means2D, "visibility_filter": None, "radii": radii,
                              "seg_feature": extra_attrs, "gau_related_pixels": None})
        if len(res) > 5:
            rets["sampled_seg_feature"] = res[5]
        dict.update(rets, dict.fromkeys(_LAZY_KEYS))
        return rets

    rets = RenderPackage({"render": rendered_image, "viewspace_points": means2D, "visibility_filter": None,
                          "radii": radii, "seg_feature": extra_attrs, "gau_related_pixels": gau_related_pixels})
    if len(res) > 5:
        # extension: ``seg_feature.reshape(F, -1)[:, sample_pixels].T`` without a dense gradient map in the backward
        rets["sampled_seg_feature"] = res[5]
    if getattr(pipe, "lazy_maps", False) or _ENV_LAZY_MAPS:
        dict.update(rets, dict.fromkeys(_LAZY_KEYS))
        rets._pending = (viewpoint_camera, allmap, pipe.depth_ratio, torch.is_grad_enabled())
    else:
        rets.update(post_process(viewpoint_camera, allmap, pipe.depth_ratio))
    return rets
