"""COLMAP sparse models -> cameras and the initial point cloud (reference scene/colmap_loader.py binary/text readers and
scene/dataset_readers.py:48-124,142-190).  Binary layouts are COLMAP's documented ones:

    cameras.bin   u64 n | n x (i32 camera_id, i32 model_id, u64 width, u64 height, f64 params[k(model)])
    images.bin    u64 n | n x (i32 image_id, f64 qvec[4], f64 tvec[3], i32 camera_id, name\\0, u64 m, m x (f64 x, f64 y, i64 point3D_id))
    points3D.bin  u64 n | n x (i64 id, f64 xyz[3], u8 rgb[3], f64 error, u64 t, t x (i32 image_id, i32 point2D_idx))

The whole file is read once and walked with ``struct.unpack_from``; per-image 2-D observations and per-point tracks —
which the training pipeline never looks at — are skipped by their byte length instead of being unpacked one record at a
time.  What the pipeline consumes is produced directly: per image ``R = qvec2rotmat(q)^T`` (camera-to-world rotation),
``T = tvec``, the two fields of view from the focal lengths, and the camera objects' matrices (``scenes.Camera``), plus
``nerfpp_norm`` (centre / 1.1 x radius of the camera positions = ``cameras_extent``).  Only undistorted models
(SIMPLE_PINHOLE, PINHOLE; SIMPLE_RADIAL and OPENCV are read with their distortion ignored, like the reference).
Writers for the three files exist for tests and for exporting synthetic scenes."""
from __future__ import annotations

import math
import os
import struct
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

MODEL_PARAMS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8),
                5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4),
                9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}
MODEL_IDS = {name: (mid, k) for mid, (name, k) in MODEL_PARAMS.items()}


@dataclass
class Intrinsics:
    id: int
    model: str
    width: int
    height: int
    params: np.ndarray


@dataclass
class Pose:
    id: int
    qvec: np.ndarray      # (w, x, y, z), world-to-camera
    tvec: np.ndarray
    camera_id: int
    name: str


@dataclass
class CameraInfo:        # the fields of the reference's CameraInfo that do not need the image file
    uid: int
    R: np.ndarray         # camera-to-world rotation (the transpose of COLMAP's)
    T: np.ndarray
    FovY: float
    FovX: float
    image_name: str
    image_path: str
    width: int
    height: int


def qvec_to_rotmat(q) -> np.ndarray:
    w, x, y, z = (float(v) for v in q)
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def focal_to_fov(focal: float, pixels: float) -> float:
    return 2 * math.atan(pixels / (2 * focal))


# ----------------------------------------------------------------------------- binary readers
def read_cameras_bin(path: str) -> Dict[int, Intrinsics]:
    buf = open(path, "rb").read()
    (n,), off, out = struct.unpack_from("<Q", buf, 0), 8, {}
    for _ in range(n):
        cid, mid, w, h = struct.unpack_from("<iiQQ", buf, off)
        off += 24
        name, k = MODEL_PARAMS[mid]
        out[cid] = Intrinsics(cid, name, int(w), int(h), np.frombuffer(buf, "<f8", k, off).copy())
        off += 8 * k
    return out


def read_images_bin(path: str) -> Dict[int, Pose]:
    buf = open(path, "rb").read()
    (n,), off, out = struct.unpack_from("<Q", buf, 0), 8, {}
    for _ in range(n):
        iid = struct.unpack_from("<i", buf, off)[0]
        qt = np.frombuffer(buf, "<f8", 7, off + 4).copy()
        cam = struct.unpack_from("<i", buf, off + 60)[0]
        end = buf.index(b"\x00", off + 64)
        name = buf[off + 64:end].decode("utf-8")
        (m,) = struct.unpack_from("<Q", buf, end + 1)
        off = end + 9 + 24 * m                    # skip the (x, y, point3D_id) observations
        out[iid] = Pose(iid, qt[:4], qt[4:], cam, name)
    return out


def read_points3d_bin(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(xyz [n,3] f64, rgb [n,3] u8, error [n,1] f64), in file order, like ``read_points3D_binary``."""
    buf = open(path, "rb").read()
    (n,), off = struct.unpack_from("<Q", buf, 0), 8
    xyz, rgb, err = np.empty((n, 3)), np.empty((n, 3), dtype=np.uint8), np.empty((n, 1))
    for i in range(n):
        xyz[i] = struct.unpack_from("<3d", buf, off + 8)
        rgb[i] = struct.unpack_from("<3B", buf, off + 32)
        err[i, 0], t = struct.unpack_from("<dQ", buf, off + 35)
        off += 51 + 8 * t                          # skip the track
    return xyz, rgb, err


# ----------------------------------------------------------------------------- text readers (cameras.txt / images.txt / points3D.txt)
def _data_lines(path):
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line and not line.startswith("#"):
                yield line


def read_cameras_txt(path: str) -> Dict[int, Intrinsics]:
    out = {}
    for line in _data_lines(path):
        e = line.split()
        out[int(e[0])] = Intrinsics(int(e[0]), e[1], int(e[2]), int(e[3]), np.array([float(v) for v in e[4:]]))
    return out


def read_images_txt(path: str) -> Dict[int, Pose]:
    out, lines = {}, list(_raw_lines(path))
    i = 0
    while i < len(lines):
        e = lines[i].split()
        out[int(e[0])] = Pose(int(e[0]), np.array([float(v) for v in e[1:5]]), np.array([float(v) for v in e[5:8]]), int(e[8]), e[9])
        i += 2                                      # the second line of a pair lists the 2-D observations (may be empty)
    return out


def _raw_lines(path):
    with open(path) as f:
        started = False
        for line in f:
            s = line.rstrip("\n")
            if not started and (not s.strip() or s.lstrip().startswith("#")):
                continue
            started = True
            yield s


def read_points3d_txt(path: str):
    rows = [line.split() for line in _data_lines(path)]
    xyz = np.array([[float(v) for v in r[1:4]] for r in rows]).reshape(-1, 3)
    rgb = np.array([[int(v) for v in r[4:7]] for r in rows], dtype=np.uint8).reshape(-1, 3)
    err = np.array([[float(r[7])] for r in rows]).reshape(-1, 1)
    return xyz, rgb, err


# ----------------------------------------------------------------------------- what the pipeline consumes
def camera_infos(poses: Dict[int, Pose], intrinsics: Dict[int, Intrinsics], images_folder: str = "images") -> List[CameraInfo]:
    """``readColmapCameras`` (dataset_readers.py:71-108) without opening the image files, sorted by image name (:163)."""
    out = []
    for pose in poses.values():
        intr = intrinsics[pose.camera_id]
        if intr.model in ("SIMPLE_PINHOLE", "SIMPLE_RADIAL"):
            fx = fy = float(intr.params[0])
        elif intr.model in ("PINHOLE", "OPENCV"):
            fx, fy = float(intr.params[0]), float(intr.params[1])
        else:
            raise ValueError(f"COLMAP camera model {intr.model}: only undistorted (PINHOLE-like) datasets are supported")
        path = os.path.join(images_folder, os.path.basename(pose.name))
        out.append(CameraInfo(intr.id, qvec_to_rotmat(pose.qvec).T, np.array(pose.tvec), focal_to_fov(fy, intr.height),
                              focal_to_fov(fx, intr.width), os.path.basename(path).split(".")[0], path, intr.width, intr.height))
    return sorted(out, key=lambda c: c.image_name)


def nerfpp_norm(infos: List[CameraInfo]) -> Dict[str, object]:
    """Centre of the camera positions and 1.1 x the largest distance to it (dataset_readers.py:48-68):
    ``radius`` is the ``cameras_extent`` densification and the learning rates are scaled with."""
    centers = np.stack([-(c.R @ c.T) for c in infos], axis=1)       # C2W[:3,3] of W2C = [R^T | T]
    center = centers.mean(axis=1, keepdims=True)
    radius = float(np.linalg.norm(centers - center, axis=0).max()) * 1.1
    return {"translate": -center.flatten(), "radius": radius}


def load_sparse_model(scene_dir: str):
    """(poses, intrinsics, (xyz, rgb, err)) from ``scene_dir`` — binary files, else the text ones (:147-156,175-180)."""
    def pick(stem, rb, rt):
        b, t = os.path.join(scene_dir, stem + ".bin"), os.path.join(scene_dir, stem + ".txt")
        return rb(b) if os.path.exists(b) else rt(t)
    return (pick("images", read_images_bin, read_images_txt), pick("cameras", read_cameras_bin, read_cameras_txt),
            pick("points3D", read_points3d_bin, read_points3d_txt))


def scene_cameras(infos: List[CameraInfo], znear: float = 0.01, zfar: float = 100.0):
    """``scenes.Camera`` objects (the matrices of scene/cameras.py:81-86) for a list of camera infos."""
    import torch
    from . import scenes
    cams = []
    for c in infos:
        w2c = np.eye(4, dtype=np.float32)
        w2c[:3, :3] = c.R.T
        w2c[:3, 3] = c.T
        wvt = torch.tensor(w2c).T.contiguous()
        proj = scenes.projection_matrix(znear, zfar, c.FovX, c.FovY).T.contiguous()
        full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        cams.append(scenes.Camera(int(c.width), int(c.height), float(c.FovX), float(c.FovY), wvt, proj, full,
                                  wvt.inverse()[3, :3].contiguous()))
    return cams


# ----------------------------------------------------------------------------- writers
def write_cameras_bin(path: str, intrinsics: Dict[int, Intrinsics]) -> None:
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(intrinsics)))
        for c in intrinsics.values():
            mid, k = MODEL_IDS[c.model]
            assert len(c.params) == k
            f.write(struct.pack("<iiQQ", c.id, mid, c.width, c.height) + np.asarray(c.params, "<f8").tobytes())


def write_images_bin(path: str, poses: Dict[int, Pose], observations: Dict[int, np.ndarray] = None) -> None:
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(poses)))
        for p in poses.values():
            f.write(struct.pack("<i", p.id) + np.asarray(p.qvec, "<f8").tobytes() + np.asarray(p.tvec, "<f8").tobytes())
            f.write(struct.pack("<i", p.camera_id) + p.name.encode("utf-8") + b"\x00")
            obs = None if observations is None else observations.get(p.id)
            m = 0 if obs is None else len(obs)
            f.write(struct.pack("<Q", m))
            for j in range(m):
                f.write(struct.pack("<ddq", float(obs[j][0]), float(obs[j][1]), int(obs[j][2])))


def write_points3d_bin(path: str, xyz, rgb, err, tracks: List[np.ndarray] = None) -> None:
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(xyz)))
        for i in range(len(xyz)):
            tr = [] if tracks is None else tracks[i]
            f.write(struct.pack("<q3d3BdQ", i + 1, *[float(v) for v in xyz[i]], *[int(v) for v in rgb[i]], float(err[i]), len(tr)))
            for a, b in tr:
                f.write(struct.pack("<ii", int(a), int(b)))
