"""Adaptive density control of the RGB/geometry stage (reference train.py:138-151; scene/gaussian_model.py:358-362,
433-605 — SURVEY §8f rank 3), for a model whose parameters live in single-tensor Adam groups named
``xyz, f_dc, f_rest, opacity, scaling, rotation`` (``harness.RgbGaussianModel``).

What it does, every iteration (``accumulate``): per visible Gaussian, add the norm of the screen-space positional gradient
to an accumulator, count the observation, keep the largest screen radius seen — one fused HIP pass over the P rows
(``iso_densify_stats``) where the reference issues a masked gather / norm / scatter and a masked max (six launches).
Every ``densification_interval`` iterations (``densify_and_prune``): clone small Gaussians and split large ones whose
average gradient is above the threshold, then prune transparent / oversized ones; ``reset_opacity`` every
``opacity_reset_interval`` iterations.  Rows are appended / removed in the parameters AND in Adam's moment estimates
(new rows start with zero moments), which is one generic row edit here (``_edit_rows``).

Reference behaviours kept on purpose: the statistics (including ``max_radii2D``) are re-zeroed by every append, so the
screen-size pruning test of the same call never fires; the split samples ``N = 2`` children from the parent's own
Gaussian with scales divided by ``0.8 N``; the gradient padded for the split treats freshly cloned rows as zero.
Pinned against the reference's own outputs in ``tests/golden/densify.npz`` (same RNG stream on the host)."""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch
import torch.nn as nn

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
_ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
         "rotation": "_rotation"}


def rotation_matrices(q: torch.Tensor) -> torch.Tensor:
    """[n,4] (w,x,y,z) quaternions, normalised here -> [n,3,3] (utils/general_utils.py:79-103)."""
    q = q / torch.sqrt(q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1] + q[:, 2] * q[:, 2] + q[:, 3] * q[:, 3])[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = (1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y))
    return torch.stack(rows, dim=1).reshape(-1, 3, 3)


class Densifier:
    def __init__(self, model, optimizer: torch.optim.Optimizer, percent_dense: float = 0.01):
        self.model, self.opt, self.percent_dense = model, optimizer, float(percent_dense)
        self._groups = {g["name"]: g for g in optimizer.param_groups}
        missing = [n for n in GROUPS if n not in self._groups]
        if missing:
            raise ValueError(f"optimizer has no parameter group named {missing}")
        P, dev = model._xyz.shape[0], model._xyz.device
        self.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
        self.denom = torch.zeros((P, 1), device=dev)
        self.max_radii2D = torch.zeros((P,), device=dev)

    # ------------------------------------------------------------------ per-iteration statistics
    def accumulate(self, viewspace_grad: torch.Tensor, visibility_filter: torch.Tensor, radii: torch.Tensor) -> None:
        """train.py:140-142: ``max_radii2D[vis] = max(max_radii2D[vis], radii[vis])`` and
        ``add_densification_stats`` (gaussian_model.py:601-604)."""
        if viewspace_grad.is_cuda:
            from ._lib import check, lib
            g = viewspace_grad.contiguous().float()
            vis = visibility_filter.contiguous()
            vis = vis.view(torch.uint8) if vis.dtype == torch.bool else vis.to(torch.uint8)
            rad = radii.contiguous().to(torch.int32)
            p = lambda t: ctypes.c_void_p(t.data_ptr())
            with torch.cuda.device(g.device):
                check(lib().iso_densify_stats(g.shape[0], g.shape[1], p(g), p(vis), p(rad), p(self.xyz_gradient_accum),
                                              p(self.denom), p(self.max_radii2D),
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "iso_densify_stats")
            return
        vis = visibility_filter.bool()
        self.max_radii2D[vis] = torch.max(self.max_radii2D[vis], radii[vis].to(self.max_radii2D.dtype))
        self.xyz_gradient_accum[vis] += torch.norm(viewspace_grad[vis], dim=-1, keepdim=True)
        self.denom[vis] += 1

    # ------------------------------------------------------------------ row edits of parameters + Adam moments
    def _edit_rows(self, keep: Optional[torch.Tensor], extra: Optional[Dict[str, torch.Tensor]]) -> None:
        """Keep the rows selected by the boolean mask ``keep`` (all if None), then append ``extra[name]`` (if given) to
        every group; Adam's ``exp_avg`` / ``exp_avg_sq`` follow (appended rows get zero moments)."""
        for name in GROUPS:
            grp = self._groups[name]
            old = grp["params"][0]
            state = self.opt.state.pop(old, None)
            data = old.detach()
            rows = data if keep is None else data[keep]
            add = None if extra is None else extra[name]
            new = nn.Parameter((rows if add is None else torch.cat((rows, add), dim=0)).requires_grad_(True))
            if state is not None and "exp_avg" in state:
                for key in ("exp_avg", "exp_avg_sq"):
                    m = state[key] if keep is None else state[key][keep]
                    state[key] = m if add is None else torch.cat((m, torch.zeros_like(add)), dim=0)
                self.opt.state[new] = state
            grp["params"][0] = new
            setattr(self.model, _ATTR[name], new)

    def _reset_stats(self) -> None:
        P, dev = self.model._xyz.shape[0], self.model._xyz.device
        self.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
        self.denom = torch.zeros((P, 1), device=dev)
        self.max_radii2D = torch.zeros((P,), device=dev)

    def _append(self, extra: Dict[str, torch.Tensor]) -> None:
        self._edit_rows(None, extra)
        self._reset_stats()                                  # gaussian_model.py:535-537

    def _prune(self, drop: torch.Tensor) -> None:
        keep = ~drop
        self._edit_rows(keep, None)
        self.xyz_gradient_accum = self.xyz_gradient_accum[keep]
        self.denom = self.denom[keep]
        self.max_radii2D = self.max_radii2D[keep]

    # ------------------------------------------------------------------ clone / split / prune
    def densify_and_prune(self, max_grad: float, min_opacity: float, extent: float, max_screen_size) -> None:
        m = self.model
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        limit = self.percent_dense * extent
        # clone: under-reconstructed regions covered by small Gaussians (gaussian_model.py:568-585)
        small = torch.exp(m._scaling).max(dim=1).values <= limit
        sel = (torch.norm(grads, dim=-1) >= max_grad) & small
        self._append({n: getattr(m, _ATTR[n]).detach()[sel] for n in GROUPS})
        # split: over-reconstructed regions covered by large ones (:539-566); clones made above count as zero gradient
        P1 = m._xyz.shape[0]
        padded = torch.zeros((P1,), device=grads.device)
        padded[:grads.shape[0]] = grads.squeeze()
        scale_act = torch.exp(m._scaling.detach())
        sel = (padded >= max_grad) & (scale_act.max(dim=1).values > limit)
        N = 2
        stds = scale_act[sel].repeat(N, 1)
        stds = torch.cat([stds, torch.zeros_like(stds[:, :1])], dim=-1)
        offsets = torch.normal(mean=torch.zeros_like(stds), std=stds)
        rots = rotation_matrices(m._rotation.detach()[sel]).repeat(N, 1, 1)
        children = {
            "xyz": torch.bmm(rots, offsets.unsqueeze(-1)).squeeze(-1) + m._xyz.detach()[sel].repeat(N, 1),
            "scaling": torch.log(scale_act[sel].repeat(N, 1) / (0.8 * N)),
            "rotation": m._rotation.detach()[sel].repeat(N, 1),
            "f_dc": m._features_dc.detach()[sel].repeat(N, 1, 1),
            "f_rest": m._features_rest.detach()[sel].repeat(N, 1, 1),
            "opacity": m._opacity.detach()[sel].repeat(N, 1),
        }
        n_children = children["xyz"].shape[0]
        self._append(children)
        self._prune(torch.cat((sel, torch.zeros(n_children, dtype=torch.bool, device=sel.device))))
        # prune: transparent, and (after the first opacity reset) oversized ones (:587-599)
        drop = (torch.sigmoid(m._opacity.detach()) < min_opacity).squeeze()
        if max_screen_size:
            big_vs = self.max_radii2D > max_screen_size
            big_ws = torch.exp(m._scaling.detach()).max(dim=1).values > 0.1 * extent
            drop = drop | big_vs | big_ws
        self._prune(drop)

    def reset_opacity(self) -> None:
        """Clamp every opacity to at most 0.01 and restart its Adam moments (gaussian_model.py:358-362, 433-447)."""
        m = self.model
        op = torch.sigmoid(m._opacity.detach())
        capped = torch.min(op, torch.ones_like(op) * 0.01)
        new_raw = torch.log(capped / (1 - capped))
        grp = self._groups["opacity"]
        old = grp["params"][0]
        state = self.opt.state.pop(old, None)
        new = nn.Parameter(new_raw.requires_grad_(True))
        if state is not None:
            state["exp_avg"] = torch.zeros_like(new_raw)
            state["exp_avg_sq"] = torch.zeros_like(new_raw)
            self.opt.state[new] = state
        grp["params"][0] = new
        m._opacity = new
