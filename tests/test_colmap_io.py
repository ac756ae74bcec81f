"""COLMAP sparse-model readers against the reference's own readers: tests/golden/colmap.npz holds three binary files
(written with this package's writers) together with what scene/colmap_loader.py, dataset_readers.getNerfppNorm and
scene/cameras.Camera of the reference made of them (tests/golden/make_goldens.py)."""
import os

import numpy as np
import pytest
import torch

from instascene_amd import colmap_io as cio


@pytest.fixture()
def model_dir(tmp_path, golden_dir):
    z = np.load(os.path.join(golden_dir, "colmap.npz"))
    for k in ("cameras", "images", "points3D"):
        open(tmp_path / (k + ".bin"), "wb").write(z["file_" + k].tobytes())
    return str(tmp_path), z


def test_binary_readers_match_reference(model_dir):
    d, z = model_dir
    poses, intr, (xyz, rgb, err) = cio.load_sparse_model(d)
    np.testing.assert_array_equal(xyz, z["xyz"])
    np.testing.assert_array_equal(rgb, z["rgb"])
    np.testing.assert_array_equal(err, z["err"])
    infos = cio.camera_infos(poses, intr, "images")
    assert [c.image_name for c in infos] == [str(s) for s in z["names"]]
    assert [c.uid for c in infos] == z["uid"].tolist()
    np.testing.assert_array_equal(np.array([[c.width, c.height] for c in infos]), z["wh"])
    np.testing.assert_allclose(np.stack([c.R for c in infos]), z["R"], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(np.stack([c.T for c in infos]), z["T"])
    np.testing.assert_allclose(np.array([[c.FovX, c.FovY] for c in infos]), z["fov"], rtol=1e-15)
    norm = cio.nerfpp_norm(infos)
    np.testing.assert_allclose(norm["translate"], z["norm_translate"], atol=1e-5)
    assert abs(norm["radius"] - float(z["norm_radius"])) < 1e-5
    cams = cio.scene_cameras(infos)
    for j, cam in enumerate(cams):
        np.testing.assert_allclose(cam.world_view_transform.numpy(), z["matrices"][j, 0], atol=1e-6)
        np.testing.assert_allclose(cam.full_proj_transform.numpy(), z["matrices"][j, 1], atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(cam.camera_center.numpy(), z[f"center{j}"], atol=1e-5)


def test_text_readers_and_writer_round_trip(tmp_path, model_dir):
    d, _ = model_dir
    poses, intr, (xyz, rgb, err) = cio.load_sparse_model(d)
    # the same model as text files
    t = tmp_path / "txt"
    t.mkdir()
    with open(t / "cameras.txt", "w") as f:
        f.write("# Camera list\n")
        for c in intr.values():
            f.write(f"{c.id} {c.model} {c.width} {c.height} " + " ".join(repr(float(v)) for v in c.params) + "\n")
    with open(t / "images.txt", "w") as f:
        f.write("# Image list with two lines of data per image\n#   second line: POINTS2D[]\n")
        for p in poses.values():
            f.write(f"{p.id} " + " ".join(repr(float(v)) for v in list(p.qvec) + list(p.tvec)) + f" {p.camera_id} {p.name}\n")
            f.write("1.0 2.0 -1\n" if p.id % 2 else "\n")
    with open(t / "points3D.txt", "w") as f:
        f.write("# 3D point list\n")
        for i in range(len(xyz)):
            f.write(f"{i + 1} " + " ".join(repr(float(v)) for v in xyz[i]) + " " + " ".join(str(int(v)) for v in rgb[i])
                    + f" {float(err[i, 0])!r} 1 2\n")
    poses2, intr2, (xyz2, rgb2, err2) = cio.load_sparse_model(str(t))
    assert sorted(poses2) == sorted(poses) and sorted(intr2) == sorted(intr)
    for k in poses:
        np.testing.assert_array_equal(poses2[k].qvec, poses[k].qvec)
        np.testing.assert_array_equal(poses2[k].tvec, poses[k].tvec)
        assert poses2[k].name == poses[k].name and poses2[k].camera_id == poses[k].camera_id
    for k in intr:
        np.testing.assert_array_equal(intr2[k].params, intr[k].params)
        assert (intr2[k].model, intr2[k].width, intr2[k].height) == (intr[k].model, intr[k].width, intr[k].height)
    np.testing.assert_array_equal(xyz2, xyz); np.testing.assert_array_equal(rgb2, rgb); np.testing.assert_array_equal(err2, err)
    # writers reproduce the golden files byte for byte (observations / tracks excepted: they are dropped by the readers)
    w = tmp_path / "w"
    w.mkdir()
    cio.write_cameras_bin(str(w / "cameras.bin"), intr)
    assert open(w / "cameras.bin", "rb").read() == open(os.path.join(d, "cameras.bin"), "rb").read()
    with pytest.raises(ValueError):
        intr_bad = {1: cio.Intrinsics(1, "OPENCV_FISHEYE", 10, 10, np.zeros(8))}
        cio.camera_infos({1: cio.Pose(1, np.array([1.0, 0, 0, 0]), np.zeros(3), 1, "a.png")}, intr_bad)
