"""``render()`` — mirror of the reference's ``gaussian_renderer.render``
(gaussian_renderer/__init__.py:20-169): same signature, same returned dict
(13 entries), built on the HIP rasterizer.  The post-processing of the 7-channel
``allmap`` (:127-167) and ``depth_to_normal`` (utils/point_utils.py:10-40) are
restated here in torch (thin elementwise work)."""
from __future__ import annotations

import ctypes
import math
import os

import torch

from . import _lib
from .contrastive import row_normalize
from . import rasterizer as _rz
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


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
        if k == "gau_related_pixels" and getattr(v, "_isr_last_index", None) is not None:
            v = _rz.slice_tracer(v)
            dict.__setitem__(self, k, v)
        return v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def __iter__(self):          # also routes dict(pkg) / {**pkg} through __getitem__
        self._materialize()
        return dict.__iter__(self)

    def items(self):
        return [(k, self[k]) for k in dict.keys(self)]

    def values(self):
        return [self[k] for k in dict.keys(self)]


_RAY_CACHE = {}
_ZEROS = {}


def _camera_rays(view, device):
    """Per-pixel ray directions and origin of utils/point_utils.py:10-27 (static per camera: cached)."""
    key = (id(view), str(device))
    hit = _RAY_CACHE.get(key)
    if hit is not None and hit[0] is view.world_view_transform:
        return hit[1], hit[2]
    c2w = (view.world_view_transform.T).inverse()
    W, H = view.image_width, view.image_height
    ndc2pix = torch.tensor([[W / 2, 0, 0, W / 2], [0, H / 2, 0, H / 2], [0, 0, 0, 1]], dtype=torch.float32,
                           device=device).T
    projection_matrix = c2w.T @ view.full_proj_transform
    intrins = (projection_matrix @ ndc2pix)[:3, :3].T
    grid_x, grid_y = torch.meshgrid(torch.arange(W, device=device).float(), torch.arange(H, device=device).float(),
                                    indexing="xy")
    points = torch.stack([grid_x, grid_y, torch.ones_like(grid_x)], dim=-1).reshape(-1, 3)
    rays_d = points @ intrins.inverse().T @ c2w[:3, :3].T
    rays_o = c2w[:3, 3]
    if len(_RAY_CACHE) > 256:
        _RAY_CACHE.clear()
    _RAY_CACHE[key] = (view.world_view_transform, rays_d, rays_o)
    return rays_d, rays_o


def depths_to_points(view, depthmap):
    """utils/point_utils.py:10-27"""
    rays_d, rays_o = _camera_rays(view, depthmap.device)
    return depthmap.reshape(-1, 1) * rays_d + rays_o


def depth_to_normal(view, depth):
    """utils/point_utils.py:30-40"""
    points = depths_to_points(view, depth).reshape(*depth.shape[1:], 3)
    output = torch.zeros_like(points)
    dx = points[2:, 1:-1] - points[:-2, 1:-1]
    dy = points[1:-1, 2:] - points[1:-1, :-2]
    output[1:-1, 1:-1, :] = torch.nn.functional.normalize(torch.linalg.cross(dx, dy, dim=-1), dim=-1)
    return output


class _RenderPost(torch.autograd.Function):
    """The seven derived maps in two HIP kernels each way (``iso_render_post_forward/backward``)."""

    @staticmethod
    def forward(ctx, allmap, viewmatrix, rays_d, rays_o, depth_ratio):
        L = _lib.lib()
        am = allmap.contiguous().float()
        _, H, W = am.shape
        # one allocation, seven [C,H,W] views
        alpha, normal, dist, surf, snorm, depth, median = torch.empty((11, H, W), dtype=torch.float32,
                                                                      device=am.device).split((1, 3, 1, 1, 3, 1, 1))
        vm = viewmatrix.contiguous().float()
        with torch.cuda.device(am.device):
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
        with torch.cuda.device(am.device):
            _lib.check(L.iso_render_post_backward(W, H, ctx.ratio, _ptr(am), _ptr(vm), _ptr(rays_d), _ptr(rays_o), _ptr(surf),
                                                  _ptr(g_alpha), _ptr(g_normal), _ptr(g_dist), _ptr(g_surf), _ptr(g_snorm),
                                                  _ptr(g_depth), _ptr(g_median), _ptr(scratch), _ptr(out), _stream()),
                       "iso_render_post_backward")
        return out, None, None, None, None


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def post_process(viewpoint_camera, allmap, depth_ratio):
    """gaussian_renderer/__init__.py:127-167.  CUDA tensors: two HIP kernels each way; the torch restatement below
    serves host-side tensors (it is what tests/golden/render_post.npz pins)."""
    if allmap.is_cuda:
        rays_d, rays_o = _camera_rays(viewpoint_camera, allmap.device)
        alpha, normal, dist, surf, snorm, depth, median = _RenderPost.apply(
            allmap, viewpoint_camera.world_view_transform, rays_d.contiguous(), rays_o.contiguous(), float(depth_ratio))
        return {'rend_alpha': alpha, 'rend_normal': normal, 'rend_dist': dist, 'surf_depth': surf,
                'surf_normal': snorm, 'rend_depth': depth, 'rend_median_depth': median}
    render_alpha = allmap[1:2]
    render_normal = allmap[2:5]
    # (n.permute(1,2,0) @ R^T).permute(2,0,1) of the reference (:132), written as three broadcast FMAs — a [N,3]x[3,3]
    # product is a poor fit for a GEMM kernel
    Rm = viewpoint_camera.world_view_transform[:3, :3]
    render_normal = (render_normal[0:1] * Rm[:, 0].view(3, 1, 1) + render_normal[1:2] * Rm[:, 1].view(3, 1, 1)
                     + render_normal[2:3] * Rm[:, 2].view(3, 1, 1))
    render_depth_median = torch.nan_to_num(allmap[5:6], 0, 0)
    render_depth_expected = torch.nan_to_num(allmap[0:1] / render_alpha, 0, 0)
    render_dist = allmap[6:7]
    surf_depth = render_depth_expected * (1 - depth_ratio) + depth_ratio * render_depth_median
    surf_normal = depth_to_normal(viewpoint_camera, surf_depth).permute(2, 0, 1)
    surf_normal = surf_normal * render_alpha.detach()
    return {'rend_alpha': render_alpha, 'rend_normal': render_normal, 'rend_dist': render_dist,
            'surf_depth': surf_depth, 'surf_normal': surf_normal, 'rend_depth': render_depth_expected,
            'rend_median_depth': render_depth_median}


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
        splat2world = pc.get_covariance(scaling_modifier)
        W, H = viewpoint_camera.image_width, viewpoint_camera.image_height
        near, far = viewpoint_camera.znear, viewpoint_camera.zfar
        ndc2pix = torch.tensor([[W / 2, 0, 0, (W - 1) / 2], [0, H / 2, 0, (H - 1) / 2], [0, 0, far - near, near],
                                [0, 0, 0, 1]], dtype=torch.float32, device=xyz.device).T
        world2pix = viewpoint_camera.full_proj_transform @ ndc2pix
        cov3D_precomp = (splat2world[:, [0, 1, 3]] @ world2pix[:, [0, 1, 3]]).permute(0, 2, 1).reshape(-1, 9)
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
           norm_seg_feat=True, sample_pixels=None):
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
    if need_geom_grad:
        screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device)
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
    else:
        # a constant: one zero tensor per (shape, device) instead of an 18 MB fill per call
        zkey = (tuple(xyz.shape), xyz.dtype, str(xyz.device))
        screenspace_points = _ZEROS.get(zkey)
        if screenspace_points is None:
            if len(_ZEROS) > 8:
                _ZEROS.clear()
            screenspace_points = _ZEROS[zkey] = torch.zeros_like(xyz, requires_grad=False)
    rasterizer, geo = _geometry_inputs(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color)
    _rz._CONFIG["lazy_tracer"] = True       # render() defers the tracer slice (and its host sync) to first access

    means2D = screenspace_points
    seg_feature = pc.get_seg_feature
    if seg_feature is not None and norm_seg_feat:
        seg_feature = row_normalize(seg_feature, 1e-9)      # reference :61-62

    res = rasterizer(means2D=means2D, extra_attrs=seg_feature, sample_pixels=sample_pixels if seg_feature is not None else None,
                     **geo)
    rendered_image, radii, allmap, extra_attrs, gau_related_pixels = res[:5]
    _rz._CONFIG["lazy_tracer"] = False

    rets = RenderPackage({"render": rendered_image, "viewspace_points": means2D, "visibility_filter": radii > 0,
                          "radii": radii, "seg_feature": extra_attrs, "gau_related_pixels": gau_related_pixels})
    if len(res) > 5:
        # extension: ``seg_feature.reshape(F, -1)[:, sample_pixels].T`` without a dense gradient map in the backward
        rets["sampled_seg_feature"] = res[5]
    if getattr(pipe, "lazy_maps", False) or os.environ.get("ISR_LAZY_MAPS", "0") == "1":
        dict.update(rets, dict.fromkeys(_LAZY_KEYS))
        rets._pending = (viewpoint_camera, allmap, pipe.depth_ratio, torch.is_grad_enabled())
    else:
        rets.update(post_process(viewpoint_camera, allmap, pipe.depth_ratio))
    return rets
