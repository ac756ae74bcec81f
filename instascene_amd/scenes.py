"""Deterministic synthetic scenes, cameras and label maps.

No datasets or checkpoints are reachable from the build/bench machines, so the
benchmark and the parity tests run on seeded synthetic inputs with the shapes
and statistics of the reference's workloads.  Camera matrices follow the
reference's conventions exactly (row-vector convention, row-major storage):

* ``world_view_transform = W2C^T``                      (scene/cameras.py:81)
* ``projection_matrix    = getProjectionMatrix(...)^T`` (utils/graphics_utils.py:51-71)
* ``full_proj_transform  = world_view_transform @ projection_matrix`` (cameras.py:84-85)
* ``camera_center        = inverse(world_view_transform)[3, :3]``     (cameras.py:86)
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional

import torch

SEED_BASE = 20250725

# name -> (P, W, H, F, mu_s)   BASELINE.json configs C1..C5
CONFIGS = {
    "C1": dict(P=50_000, W=256, H=256, F=0, mu_s=math.log(0.024), index=1),
    "C2": dict(P=300_000, W=779, H=519, F=0, mu_s=math.log(0.0064), index=2),
    "C3": dict(P=1_500_000, W=1920, H=1080, F=32, mu_s=math.log(0.0064), index=3),
    "C4": dict(P=1_500_000, W=1920, H=1080, F=32, mu_s=math.log(0.0064), index=3),
    "C5": dict(P=5_000_000, W=1296, H=968, F=64, mu_s=math.log(0.0064), index=5),
}


@dataclass
class Camera:
    """Minimal stand-in for ``scene.cameras.Camera`` — only the attributes that
    ``gaussian_renderer.render`` and ``utils.point_utils`` read."""
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor   # [4,4] = W2C^T
    projection_matrix: torch.Tensor      # [4,4] = P^T
    full_proj_transform: torch.Tensor    # [4,4]
    camera_center: torch.Tensor          # [3]
    znear: float = 0.01
    zfar: float = 100.0
    segmap: Optional[torch.Tensor] = None
    sorted_segmap: Optional[torch.Tensor] = None
    image_name: str = ""

    def to(self, device):
        for k in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center",
                  "segmap", "sorted_segmap"):
            v = getattr(self, k)
            if v is not None:
                setattr(self, k, v.to(device))
        return self


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> torch.Tensor:
    """utils/graphics_utils.py:51-71 (returns P, not P^T)."""
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera_from_w2c(w2c: torch.Tensor, fovx: float, fovy: float, W: int, H: int, znear=0.01, zfar=100.0,
                    name="") -> Camera:
    w2c = w2c.to(torch.float32)
    wvt = w2c.transpose(0, 1).contiguous()
    proj = projection_matrix(znear, zfar, fovx, fovy).transpose(0, 1).contiguous()
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wvt.inverse()[3, :3].contiguous()
    return Camera(W, H, fovx, fovy, wvt, proj, full, center, znear, zfar, image_name=name)


def look_at_w2c(eye: torch.Tensor, target: torch.Tensor, up=(0.0, 1.0, 0.0)) -> torch.Tensor:
    """World-to-camera with +z forward, +x right, +y down (COLMAP/3DGS convention)."""
    eye = eye.to(torch.float64)
    fwd = target.to(torch.float64) - eye
    fwd = fwd / fwd.norm()
    upv = torch.tensor(up, dtype=torch.float64)
    right = torch.linalg.cross(fwd, upv)
    right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    Rm = torch.stack([right, down, fwd], dim=0)       # rows = camera axes in world coords
    w2c = torch.eye(4, dtype=torch.float64)
    w2c[:3, :3] = Rm
    w2c[:3, 3] = -Rm @ eye
    return w2c.to(torch.float32)


def ring_cameras(n: int, W: int, H: int, radius=4.0, height=0.5, fovx_deg=60.0) -> List[Camera]:
    fovx = math.radians(fovx_deg)
    fovy = 2.0 * math.atan(math.tan(fovx / 2) * H / W)
    cams = []
    for i in range(n):
        a = 2.0 * math.pi * i / n
        eye = torch.tensor([radius * math.cos(a), height * math.sin(3.0 * a), radius * math.sin(a)])
        cams.append(camera_from_w2c(look_at_w2c(eye, torch.zeros(3)), fovx, fovy, W, H, name=f"ring_{i:03d}"))
    return cams


@dataclass
class Scene:
    xyz: torch.Tensor          # [P,3]
    log_scale: torch.Tensor    # [P,2]   (pre-activation, exp -> scale)
    rot: torch.Tensor          # [P,4]   (pre-activation, normalise -> wxyz quaternion)
    opacity_logit: torch.Tensor  # [P,1]
    features_dc: torch.Tensor  # [P,1,3]
    features_rest: torch.Tensor  # [P,15,3]
    seg_feature: Optional[torch.Tensor]  # [P,F] or None
    labels3d: Optional[torch.Tensor]  # [P] int64

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device))
        return self


def morton_order(xyz: torch.Tensor, bits: int = 10) -> torch.Tensor:
    """Permutation that sorts points along the Z-order curve of their bounding box (``bits`` per axis)."""
    lo, hi = xyz.min(dim=0).values, xyz.max(dim=0).values
    q = ((xyz - lo) / (hi - lo).clamp_min(1e-12) * ((1 << bits) - 1)).round().clamp(0, (1 << bits) - 1).to(torch.int64)
    code = torch.zeros(xyz.shape[0], dtype=torch.int64, device=xyz.device)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return torch.argsort(code, stable=True)


def spatially_sorted(scene: "Scene", perm: Optional[torch.Tensor] = None) -> "Scene":
    """The same Gaussians stored in Z-order of their centres.  Every per-view result is a per-Gaussian or per-pixel
    quantity, so only the row order of ``[P, ...]`` tensors changes; neighbours in memory become neighbours on screen,
    which is what the binning scatter, the record gathers and the per-Gaussian gradient rows want."""
    if perm is None:
        perm = morton_order(scene.xyz)
    pick = lambda t: None if t is None else t[perm.to(t.device)].contiguous()
    return Scene(pick(scene.xyz), pick(scene.log_scale), pick(scene.rot), pick(scene.opacity_logit), pick(scene.features_dc),
                 pick(scene.features_rest), pick(scene.seg_feature), pick(scene.labels3d))


def synthetic_scene(P: int, F: int, seed: int, mu_s: float, extent: float = 1.5, n_labels: int = 64) -> Scene:
    g = torch.Generator(device="cpu").manual_seed(seed)
    xyz = (torch.rand(P, 3, generator=g) * 2 - 1) * extent
    log_scale = mu_s + 0.4 * torch.randn(P, 2, generator=g)
    rot = torch.randn(P, 4, generator=g)
    opacity_logit = 1.5 * torch.randn(P, 1, generator=g)
    dc = torch.randn(P, 1, 3, generator=g)
    rest = 0.1 * torch.randn(P, 15, 3, generator=g)
    feat = None
    if F > 0:
        feat = torch.randn(P, F, generator=g)
        feat = feat / feat.norm(dim=1, keepdim=True)
    labels = torch.randint(0, n_labels, (P,), generator=g)
    return Scene(xyz, log_scale, rot, opacity_logit, dc, rest, feat, labels)


def config_scene(name: str, scale: float = 1.0) -> tuple:
    """(scene, cameras, cfg) for a BASELINE config; ``scale``<1 shrinks P and the
    image proportionally (used for CPU-baseline samples)."""
    cfg = dict(CONFIGS[name])
    if scale != 1.0:
        cfg["P"] = max(1, int(cfg["P"] * scale * scale))
        cfg["W"] = max(16, int(cfg["W"] * scale))
        cfg["H"] = max(16, int(cfg["H"] * scale))
        # keep the projected footprint in pixels: world scale shrinks with resolution
        cfg["mu_s"] = cfg["mu_s"]
    scene = synthetic_scene(cfg["P"], cfg["F"], SEED_BASE + cfg["index"], cfg["mu_s"])
    cams = ring_cameras(64, cfg["W"], cfg["H"])
    return scene, cams, cfg


def voronoi_labels(W: int, H: int, K: int, seed: int, zero_frac: float = 0.1, device="cpu") -> torch.Tensor:
    """Voronoi partition of the image by K seeded random sites; ``zero_frac`` of
    the sites carry label 0 (= unlabeled, like the reference's filtered masks)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sites = (torch.rand(K, 2, generator=g) * torch.tensor([W, H], dtype=torch.float32)).to(device)
    lab = torch.arange(1, K + 1)
    lab[torch.rand(K, generator=g) < zero_frac] = 0
    lab = lab.to(device)
    ys = torch.arange(H, dtype=torch.float32, device=device).view(1, H, 1)
    xs = torch.arange(W, dtype=torch.float32, device=device).view(1, 1, W)
    best = torch.full((H, W), float("inf"), device=device)
    out = torch.zeros(H, W, dtype=torch.int64, device=device)
    # nearest site, 16 sites at a time (a few launches per map instead of five per site); the earlier site wins a tie
    for k0 in range(0, K, 16):
        sx = sites[k0:k0 + 16, 0].view(-1, 1, 1)
        sy = sites[k0:k0 + 16, 1].view(-1, 1, 1)
        d, arg = ((xs - sx) ** 2 + (ys - sy) ** 2).min(dim=0)
        m = d < best
        best = torch.where(m, d, best)
        out = torch.where(m, lab[k0:k0 + 16][arg], out)
    return out


def activated_inputs(scene: Scene):
    """The reference's getters (scene/gaussian_model.py:109-138) as plain tensors."""
    scales = torch.exp(scene.log_scale)
    rots = torch.nn.functional.normalize(scene.rot)
    opac = torch.sigmoid(scene.opacity_logit)
    shs = torch.cat((scene.features_dc, scene.features_rest), dim=1)
    feat = None
    if scene.seg_feature is not None:
        feat = scene.seg_feature / (scene.seg_feature.norm(dim=1, keepdim=True) + 1e-6)
        feat = feat / (feat.norm(dim=-1, keepdim=True) + 1e-9)
    return dict(means3D=scene.xyz, scales=scales, rotations=rots, opacities=opac, shs=shs, extra=feat)
