"""``GaussianAdam`` - the optimiser of the reference's ``train.py`` loop (``torch.optim.Adam(l, lr=0.0, eps=1e-15)`` over the
six parameter groups of ``GaussianModel.training_setup``, scene/gaussian_model.py:206-253; stepped at train.py:153-156) as ONE
kernel per iteration (``iso_gaussian_adam_step``, csrc/iso_optim.hip) that also

* applies the chain rule from the gradients of the ACTIVATED tensors the rasterizer consumed - ``exp(_scaling)``,
  ``sigmoid(_opacity)``, ``normalize(_rotation)``, ``cat(_features_dc, _features_rest)`` (the model's getters,
  scene/gaussian_model.py:109-138) - to the raw parameters, and
* emits those activations of the updated parameters for the next forward.

The activations are handed to ``render()`` as autograd LEAVES (``begin()``), so a step is: ``begin()`` -> ``render`` ->
loss -> ``backward()`` -> ``step()``.  Arithmetic and semantics are dense ``torch.optim.Adam``'s (a Gaussian that was not
visible still decays its moments and moves by its momentum); ``tests/test_gpu_harness.py`` pins it against torch's
optimiser.  Row edits by the density control (clone / split / prune) are not supported here: ``RgbTrainer(densify=...)``
keeps ``torch.optim.Adam``.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import _hot
from ._lib import check, lib

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
_ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
         "rotation": "_rotation"}


def _table(tensors):
    arr = (ctypes.c_void_p * 6)()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


class GaussianAdam:
    def __init__(self, model, lrs: Dict[str, float], betas=(0.9, 0.999), eps: float = 1e-15):
        self.model = model
        self.betas, self.eps = betas, float(eps)
        # torch-style groups, so that a learning-rate schedule (GaussianModel.update_learning_rate sets the "xyz" group's
        # lr every iteration, scene/gaussian_model.py:255-262) works unchanged
        self.param_groups = [{"params": [getattr(model, _ATTR[g])], "lr": float(lrs[g]), "name": g} for g in GROUPS]
        for g in self.param_groups:
            p = g["params"][0]
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("GaussianAdam needs contiguous float32 CUDA parameters")
        self.exp_avg = {g: torch.zeros_like(getattr(model, _ATTR[g])) for g in GROUPS}
        self.exp_avg_sq = {g: torch.zeros_like(getattr(model, _ATTR[g])) for g in GROUPS}
        self.step_count = 0
        self._acts = None               # activations of the current parameters, written by the last step
        self.leaves = None

    # -- activations --------------------------------------------------------------------------------------------------
    def _params(self):
        return [getattr(self.model, _ATTR[g]) for g in GROUPS]

    def _launch(self, grads, want_acts: bool):
        ps = self._params()
        P = ps[0].shape[0]
        M = 1 + ps[2].shape[1]
        dev = ps[0].device
        acts = None
        if want_acts:
            new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
            acts = (new(P, M, 3), new(P, 1), new(P, 2), new(P, 4))
        lr = (ctypes.c_double * 6)(*[float(g["lr"]) for g in self.param_groups])
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        g = [None if t is None else t.contiguous().float() for t in grads]
        with _hot.on_device(dev):
            check(lib().iso_gaussian_adam_step(
                P, M, _table([t.data for t in ps]), _table([self.exp_avg[k] for k in GROUPS]),
                _table([self.exp_avg_sq[k] for k in GROUPS]), lr, float(self.betas[0]), float(self.betas[1]), self.eps,
                max(1, self.step_count), p(g[0]), p(g[1]), p(g[2]), p(g[3]), p(g[4]),
                *(p(a) for a in (acts if acts is not None else (None,) * 4)),
                _hot.stream_ptr(dev)), "iso_gaussian_adam_step")
        return acts

    def begin(self):
        """The activated tensors of the current parameters as leaves that require grad:
        ``dict(xyz, shs, opacity, scaling, rotation)`` - what the model's getters return during the step."""
        versions = tuple(p._version for p in self._params())
        if self._acts is None or self._acts[0] != versions:
            self._acts = (versions, self._launch((None,) * 5, want_acts=True))     # no gradient: only the activations are written
        shs, opa, scale, rot = self._acts[1]
        mk = lambda t: t.detach().requires_grad_(True)
        self.leaves = dict(xyz=self.model._xyz, shs=mk(shs), opacity=mk(opa), scaling=mk(scale), rotation=mk(rot))
        return self.leaves

    def leaf_grads(self):
        """The five gradient tensors of the leaves (None where nothing arrived) - what a data-parallel trainer sums."""
        lv = self.leaves
        return [lv["xyz"].grad, lv["shs"].grad, lv["opacity"].grad, lv["scaling"].grad, lv["rotation"].grad]

    def step(self, grads=None):
        """Chain rule + Adam + the next activations from the leaves' gradients (or ``grads``, e.g. their sum over ranks)."""
        if self.leaves is None:
            raise RuntimeError("GaussianAdam.step: begin() was not called")
        grads = self.leaf_grads() if grads is None else grads
        self.step_count += 1
        acts = self._launch(grads, want_acts=True)
        ps = self._params()
        for prm in ps:
            torch.autograd.graph.increment_version(prm)         # the kernel wrote through raw pointers
        self._acts = (tuple(prm._version for prm in ps), acts)
        self.leaves = None

    def zero_grad(self, set_to_none: bool = True):
        for prm in self._params():
            if set_to_none:
                prm.grad = None
            elif prm.grad is not None:
                prm.grad.zero_()

    # -- torch-compatible state (checkpoints: train.py:157-159 saves optimizer.state_dict()) -------------------------------
    def state_dict(self):
        return {"state": {i: {"step": torch.tensor(float(self.step_count)), "exp_avg": self.exp_avg[g],
                              "exp_avg_sq": self.exp_avg_sq[g]} for i, g in enumerate(GROUPS)},
                "param_groups": [{"lr": pg["lr"], "name": pg["name"], "betas": self.betas, "eps": self.eps, "params": [i]}
                                 for i, pg in enumerate(self.param_groups)]}

    def load_state_dict(self, sd):
        by_name = {pg["name"]: pg for pg in sd["param_groups"]}
        for i, g in enumerate(GROUPS):
            pg = by_name[g]
            st = sd["state"].get(pg["params"][0])
            self.param_groups[i]["lr"] = float(pg["lr"])
            if st is not None:
                self.exp_avg[g].copy_(st["exp_avg"])
                self.exp_avg_sq[g].copy_(st["exp_avg_sq"])
                self.step_count = int(float(st["step"]))
        self._acts = None
