"""ORACLE — TEST INFRASTRUCTURE ONLY.

Independent, differentiable PyTorch (CPU, float64-capable) restatement of the
reference forward (forward.cu:75-145 per Gaussian, forward.cu:307-461 per
pixel).  Its autograd gradients are an *independent* derivation against which
the hand-derived backward of the C++ oracle (and therefore of the HIP kernels)
is checked — the reference's backward.cu is itself hand-derived.

The blend is evaluated densely per tile ([pixels x list] tensors + cumprod)
instead of a serial walk, so it shares no code structure with either the C++
oracle or the kernels.  Integer decisions (tile lists, sort order) are taken
from the C++ oracle's state: they are not differentiable.

Where the reference's backward deliberately differs from the true derivative
the restatement encodes the same convention, so autograd reproduces it:
  * alpha clamp min(0.99, .) is treated as identity in the backward
    (backward.cu:321,417) -> straight-through clamp here;
  * the quaternion gradient is taken w.r.t. the normalised components
    (auxiliary.h:239-283) -> callers pass unit quaternions as leaves and this
    file does not re-normalise through autograd.
"""
from __future__ import annotations

import math

import torch

NEAR, FAR = 0.2, 100.0
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def quat_to_rot(q):
    """q [P,4] (w,x,y,z), assumed unit; returns R [P,3,3] (row, col)."""
    w, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


def eval_sh_rgb(deg, shs, dirs):
    """shs [P,16,3], dirs [P,3] unit -> rgb [P,3] (before +0.5 / clamp)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = C0 * shs[:, 0]
    if deg > 0:
        res = res - C1 * y * shs[:, 1] + C1 * z * shs[:, 2] - C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * shs[:, 4] + C2[1] * yz * shs[:, 5] + C2[2] * (2 * zz - xx - yy) * shs[:, 6]
               + C2[3] * xz * shs[:, 7] + C2[4] * (xx - yy) * shs[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * shs[:, 9] + C3[1] * xy * z * shs[:, 10]
               + C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
               + C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + C3[5] * z * (xx - yy) * shs[:, 14]
               + C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return res


def per_gaussian(means3D, scales, quats, shs, view, proj, campos, W, H, sh_degree, scale_modifier=1.0):
    """K1 in torch: returns T [P,3,3] with rows (Tu,Tv,Tw), normal [P,3] (view
    space, flipped towards the camera), centre [P,2], rgb [P,3]."""
    dt = means3D.dtype
    view = view.to(dt).reshape(4, 4)
    proj = proj.to(dt).reshape(4, 4)
    P = means3D.shape[0]
    R = quat_to_rot(quats)
    L0 = R[:, :, 0] * (scale_modifier * scales[:, 0:1])
    L1 = R[:, :, 1] * (scale_modifier * scales[:, 1:2])
    L2 = R[:, :, 2]
    zeros, ones = torch.zeros(P, 1, dtype=dt), torch.ones(P, 1, dtype=dt)
    S = torch.stack([torch.cat([L0, zeros], 1), torch.cat([L1, zeros], 1), torch.cat([means3D, ones], 1)], dim=1)
    n2p = torch.tensor([[W / 2, 0, 0], [0, H / 2, 0], [0, 0, 0], [(W - 1) / 2, (H - 1) / 2, 1]], dtype=dt)
    Tm = S @ proj @ n2p            # [P, r, c] : row r of splat2world^T, pixel-coordinate c
    T = Tm.transpose(1, 2)         # rows = Tu, Tv, Tw
    normal = L2 @ view[:3, :3]     # row-vector convention: v' = v @ M[:3,:3]
    p_view = means3D @ view[:3, :3] + view[3, :3]
    cosv = -(p_view * normal).sum(-1, keepdim=True)
    normal = torch.where(cosv > 0, normal, -normal)
    Tu, Tv, Tw = T[:, 0], T[:, 1], T[:, 2]
    t = torch.tensor([9.0, 9.0, -1.0], dtype=dt)
    d = (t * Tw * Tw).sum(-1, keepdim=True)
    f = t / d
    centre = torch.stack([(f * Tu * Tw).sum(-1), (f * Tv * Tw).sum(-1)], dim=-1)
    rgb = None
    if shs is not None:
        dirs = means3D - campos.to(dt)
        dirs = dirs / dirs.norm(dim=-1, keepdim=True)
        rgb = torch.clamp_min(eval_sh_rgb(sh_degree, shs, dirs) + 0.5, 0.0)
    return T, normal, centre, rgb, p_view[:, 2]


def _st_min(a, cap):
    """min(a, cap) with identity gradient (the reference ignores the clamp in backward)."""
    return a + (torch.clamp(a, max=cap) - a).detach()


def blend(T, normal, centre, opacity, rgb, extra, bg, W, H, ranges, point_list):
    """K8 in torch, dense per tile.  Returns color [3,H,W], others [7,H,W], extra [F,H,W]."""
    dt = T.dtype
    F = 0 if extra is None else extra.shape[1]
    gx = (W + 15) // 16
    bgt = bg.to(dt)
    colors_o = []
    for tile in range(ranges.shape[0]):
        r0, r1 = int(ranges[tile, 0]), int(ranges[tile, 1])
        tx, ty = tile % gx, tile // gx
        xs = torch.arange(tx * 16, min(tx * 16 + 16, W))
        ys = torch.arange(ty * 16, min(ty * 16 + 16, H))
        if len(xs) == 0 or len(ys) == 0:
            continue
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        px = xx.reshape(-1, 1).to(dt)
        py = yy.reshape(-1, 1).to(dt)
        npx = px.shape[0]
        if r1 == r0:
            colors_o.append((tile, bgt.reshape(3, 1).expand(3, npx), torch.zeros(7, npx, dtype=dt),
                             torch.zeros(F, npx, dtype=dt), yy, xx))
            continue
        ids = torch.as_tensor(point_list[r0:r1].astype("int64"))
        Tu, Tv, Tw = T[ids, 0], T[ids, 1], T[ids, 2]         # [L,3]
        k = px.unsqueeze(-1) * Tw.unsqueeze(0) - Tu.unsqueeze(0)   # [npx,L,3]
        l = py.unsqueeze(-1) * Tw.unsqueeze(0) - Tv.unsqueeze(0)
        p = torch.linalg.cross(k, l, dim=-1)
        pz = p[..., 2]
        ok = pz != 0
        pz_safe = torch.where(ok, pz, torch.ones_like(pz))
        sx, sy = p[..., 0] / pz_safe, p[..., 1] / pz_safe
        rho3d = sx * sx + sy * sy
        dx = centre[ids, 0].unsqueeze(0) - px
        dy = centre[ids, 1].unsqueeze(0) - py
        rho2d = 2.0 * (dx * dx + dy * dy)
        use3d = (rho3d <= rho2d).detach()
        rho = torch.where(use3d, rho3d, rho2d)
        depth = torch.where(use3d, sx * Tw[:, 0] + sy * Tw[:, 1] + Tw[:, 2], Tw[:, 2].unsqueeze(0).expand_as(sx))
        ok = ok & (depth >= NEAR)
        power = -0.5 * rho
        ok = ok & (power <= 0)
        G = torch.exp(torch.where(ok, power, torch.zeros_like(power)))
        alpha = _st_min(opacity[ids].reshape(1, -1) * G, 0.99)
        ok = ok & (alpha >= 1.0 / 255.0).detach()
        alpha = torch.where(ok, alpha, torch.zeros_like(alpha))
        one_m = 1 - alpha
        Tincl = torch.cumprod(one_m, dim=1)                      # T after each entry
        Texcl = torch.cat([torch.ones(npx, 1, dtype=dt), Tincl[:, :-1]], dim=1)
        # termination: first entry with alpha>0 whose test_T < 1e-4 stops the pixel
        stop = (ok & (Tincl < 1e-4)).detach()
        alive = (torch.cumsum(stop.to(torch.int64), dim=1) == 0)
        alpha = torch.where(alive, alpha, torch.zeros_like(alpha))
        okb = ok & alive
        Tincl = torch.cumprod(1 - alpha, dim=1)
        Texcl = torch.cat([torch.ones(npx, 1, dtype=dt), Tincl[:, :-1]], dim=1)
        w = alpha * Texcl
        Tfin = Tincl[:, -1]
        m = (FAR / (FAR - NEAR)) * (1 - NEAR / torch.where(okb, depth, torch.ones_like(depth)))
        m = torch.where(okb, m, torch.zeros_like(m))
        A = 1 - Texcl
        mw, mmw = m * w, m * m * w
        M1e = torch.cumsum(mw, 1) - mw
        M2e = torch.cumsum(mmw, 1) - mmw
        dist = ((m * m * A + M2e - 2 * m * M1e) * w).sum(1)
        D = (torch.where(okb, depth, torch.zeros_like(depth)) * w).sum(1)
        med_mask = (okb & (Texcl > 0.5)).detach()
        idx = torch.arange(med_mask.shape[1]).unsqueeze(0).expand_as(med_mask)
        last = torch.where(med_mask, idx, torch.full_like(idx, -1)).max(dim=1).values
        has = last >= 0
        med = torch.where(has, depth.gather(1, last.clamp(min=0).unsqueeze(1)).squeeze(1), torch.zeros(npx, dtype=dt))
        Nn = w @ normal[ids]
        C = w @ rgb[ids] + Tfin.unsqueeze(1) * bgt.unsqueeze(0)
        E = w @ extra[ids] if F else torch.zeros(npx, 0, dtype=dt)
        oth = torch.stack([D, 1 - Tfin, Nn[:, 0], Nn[:, 1], Nn[:, 2], med, dist], dim=0)
        colors_o.append((tile, C.t(), oth, E.t(), yy, xx))
    # assemble without in-place ops on graph tensors
    color_parts = torch.zeros(3, H * W, dtype=dt)
    others_parts = torch.zeros(7, H * W, dtype=dt)
    feat_parts = torch.zeros(F, H * W, dtype=dt)
    lin_all, Cs, Os, Es = [], [], [], []
    for tile, C, oth, E, yy, xx in colors_o:
        lin_all.append((yy * W + xx).reshape(-1))
        Cs.append(C)
        Os.append(oth)
        Es.append(E)
    lin = torch.cat(lin_all)
    color_parts = color_parts.index_copy(1, lin, torch.cat(Cs, dim=1))
    others_parts = others_parts.index_copy(1, lin, torch.cat(Os, dim=1))
    if F:
        feat_parts = feat_parts.index_copy(1, lin, torch.cat(Es, dim=1))
    return color_parts.reshape(3, H, W), others_parts.reshape(7, H, W), feat_parts.reshape(F, H, W)
