"""Point-cloud checkpoints in the reference's PLY layout (scene/gaussian_model.py:263-321 ``save_ply`` /
``save_ply_as_3dgs``, :364-422 ``load_ply``), without the ``plyfile`` package: one vertex element of
little-endian float32 properties

    x y z  nx ny nz  f_dc_0..2  f_rest_0..44  opacity  scale_0..1[2]  rot_0..3  [segfeat_0..F-1]

with the SH coefficients stored channel-major (``[P,15,3]`` is written as ``[P,3,15]`` flattened, :276-278) and all
values pre-activation.  Written and parsed as one structured numpy array (the reference builds a Python tuple per
Gaussian, :289).  ``load_ply`` also reads ASCII files and other scalar property types, and ignores properties it does
not know.  The coloured preview clouds the reference writes next to the checkpoint through open3d are not produced."""
from __future__ import annotations

import math
import os
from typing import List, Optional

import numpy as np
import torch

from .scenes import Scene

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def attribute_names(n_dc: int, n_rest: int, n_scale: int, n_rot: int, n_seg: int, export_as_3dgs: bool = False) -> List[str]:
    """Property order of the reference (``construct_list_of_attributes``, :263-283)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale + (1 if export_as_3dgs else 0))]
    names += [f"rot_{i}" for i in range(n_rot)]
    names += [f"segfeat_{i}" for i in range(n_seg)]
    return names


def save_ply(path: str, xyz, features_dc, features_rest, opacity, scaling, rotation, seg_feature=None, crop_mask=None,
             export_as_3dgs: bool = False) -> None:
    """Write the raw (pre-activation) parameters.  ``export_as_3dgs`` appends a third log-scale of ``log(1e-6)`` like
    ``save_ply_as_3dgs`` (:311).  ``crop_mask``: optional boolean row selection (:266-269)."""
    def host(t):
        return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)

    keep = slice(None) if crop_mask is None else host(crop_mask).astype(bool)
    xyz_n = host(xyz)[keep].astype(np.float32)
    f_dc = host(features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous())[keep]
    f_rest = host(features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous())[keep]
    scale = host(scaling)[keep]
    if export_as_3dgs:
        scale = np.concatenate([scale, np.full_like(scale[:, :1], math.log(1e-6))], axis=-1)
    cols = [xyz_n, np.zeros_like(xyz_n), f_dc, f_rest, host(opacity)[keep].reshape(len(xyz_n), -1), scale, host(rotation)[keep]]
    n_seg = 0
    if seg_feature is not None:
        cols.append(host(seg_feature)[keep])
        n_seg = cols[-1].shape[1]
    names = attribute_names(f_dc.shape[1], f_rest.shape[1], host(scaling).shape[1], cols[6].shape[1], n_seg, export_as_3dgs)
    table = np.ascontiguousarray(np.concatenate([c.astype(np.float32) for c in cols], axis=1))
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % table.shape[0]
    header += "".join("property float %s\n" % n for n in names) + "end_header\n"
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.astype("<f4").tobytes())


def read_vertex_table(path: str):
    """``(names, {name: float64 column})`` of the first ``vertex`` element of a PLY file."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex, seen_vertex = None, 0, [], False, False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex" and not seen_vertex
                if in_vertex:
                    count, seen_vertex = int(tok[2]), True
                elif not seen_vertex:
                    raise ValueError(f"{path}: element '{tok[1]}' before the vertex element is not supported")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties in the vertex element are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2) if count else np.zeros((0, len(props)))
            cols = {n: rows[:, i].astype(np.float64) for i, (n, _) in enumerate(props)}
        elif fmt in ("binary_little_endian", "binary_big_endian"):
            order = "<" if fmt == "binary_little_endian" else ">"
            dt = np.dtype([(n, order + t) for n, t in props])
            rec = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
            cols = {n: rec[n].astype(np.float64) for n, _ in props}
        else:
            raise ValueError(f"{path}: unknown PLY format {fmt!r}")
    return [n for n, _ in props], cols


def load_ply(path: str, max_sh_degree: int = 3, seg_feat_dim: Optional[int] = None) -> Scene:
    """The reference's ``load_ply`` (:364-422) as a :class:`Scene` of host tensors.  ``seg_feat_dim``: load
    ``segfeat_*`` if exactly that many are present (None = whatever is there); 3DGS exports keep their first two scales."""
    names, col = read_vertex_table(path)
    P = len(col["x"])
    by_index = lambda prefix: sorted((n for n in names if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))
    stack = lambda ns: np.stack([col[n] for n in ns], axis=1) if ns else np.zeros((P, 0))
    xyz = stack(["x", "y", "z"])
    dc = stack(["f_dc_0", "f_dc_1", "f_dc_2"]).reshape(P, 3, 1)
    rest_names = by_index("f_rest_")
    n_rest = (max_sh_degree + 1) ** 2 - 1
    if len(rest_names) != 3 * n_rest:
        raise ValueError(f"{path}: {len(rest_names)} f_rest properties, expected {3 * n_rest} for SH degree {max_sh_degree}")
    rest = stack(rest_names).reshape(P, 3, n_rest)
    scales = stack(by_index("scale_")[:2])
    rots = stack(by_index("rot"))
    seg = None
    seg_names = by_index("segfeat")
    if seg_names and (seg_feat_dim is None or len(seg_names) == seg_feat_dim):
        seg = torch.tensor(stack(seg_names), dtype=torch.float32)
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    return Scene(t(xyz), t(scales), t(rots), t(col["opacity"][:, None]), t(dc).transpose(1, 2).contiguous(),
                 t(rest).transpose(1, 2).contiguous(), seg, None)
