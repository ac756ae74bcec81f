"""PLY checkpoints in the reference's layout (scene/gaussian_model.py:263-321, 364-422).  `plyfile` is not installed in the
build image; since round 3 the reference's own ``save_ply`` / ``load_ply`` pin the CONTENT anyway - they were run with
recording stand-ins for plyfile's two entry points, and tests/golden/ply.npz holds the vertex element exactly as the reference
builds it (property order, channel-major SH flattening, crop mask) and the tensors its loader builds back
(``test_checkpoint_equals_what_the_reference_hands_to_plyfile``).  Only the container syntax - the header lines plyfile emits
for float properties, little-endian packing: the public PLY format - is pinned by text here.  Plus round trips incl. the 3DGS
export and an ASCII file."""
import math
import struct

import numpy as np
import pytest
import torch

from instascene_amd import ply_io, scenes


def _scene(P=37, F=6, seed=3):
    return scenes.synthetic_scene(P, F, seed, math.log(0.05))


def test_header_and_record_layout(tmp_path):
    sc = _scene()
    path = str(tmp_path / "a" / "point_cloud.ply")
    ply_io.save_ply(path, sc.xyz, sc.features_dc, sc.features_rest, sc.opacity_logit, sc.log_scale, sc.rot, sc.seg_feature)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    names = [l.split()[-1] for l in lines[3:]]
    assert all(l.startswith("property float ") for l in lines[3:])
    assert names == (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)]
                     + ["opacity", "scale_0", "scale_1"] + [f"rot_{i}" for i in range(4)] + [f"segfeat_{i}" for i in range(6)])
    assert len(body) == 37 * len(names) * 4
    row0 = struct.unpack("<%df" % len(names), body[:len(names) * 4])
    np.testing.assert_array_equal(row0[0:3], sc.xyz[0].numpy())
    assert row0[3:6] == (0.0, 0.0, 0.0)
    # SH rest is channel-major on disk: f_rest_k = features_rest[:, k % 15, k // 15]
    np.testing.assert_array_equal(row0[9:9 + 45], sc.features_rest[0].T.reshape(-1).numpy())
    np.testing.assert_array_equal(row0[6:9], sc.features_dc[0, 0].numpy())


@pytest.mark.parametrize("F", [0, 6])
def test_round_trip(tmp_path, F):
    sc = _scene(F=F)
    path = str(tmp_path / "pc.ply")
    ply_io.save_ply(path, sc.xyz, sc.features_dc, sc.features_rest, sc.opacity_logit, sc.log_scale, sc.rot, sc.seg_feature)
    back = ply_io.load_ply(path)
    for a, b in [(sc.xyz, back.xyz), (sc.log_scale, back.log_scale), (sc.rot, back.rot), (sc.opacity_logit, back.opacity_logit),
                 (sc.features_dc, back.features_dc), (sc.features_rest, back.features_rest)]:
        assert torch.equal(a, b)
    assert (back.seg_feature is None) == (F == 0)
    if F:
        assert torch.equal(sc.seg_feature, back.seg_feature)
        assert ply_io.load_ply(path, seg_feat_dim=F + 1).seg_feature is None      # dimension mismatch: not loaded (:401-403)


def test_crop_mask_3dgs_export_and_ascii(tmp_path):
    sc = _scene()
    mask = torch.arange(37) % 3 == 0
    path = str(tmp_path / "crop.ply")
    ply_io.save_ply(path, sc.xyz, sc.features_dc, sc.features_rest, sc.opacity_logit, sc.log_scale, sc.rot, None,
                    crop_mask=mask, export_as_3dgs=True)
    names, col = ply_io.read_vertex_table(path)
    assert "scale_2" in names and len(col["x"]) == int(mask.sum())
    np.testing.assert_allclose(col["scale_2"], math.log(1e-6), rtol=1e-6)
    back = ply_io.load_ply(path)
    assert back.log_scale.shape == (int(mask.sum()), 2) and torch.equal(back.xyz, sc.xyz[mask])
    # an ASCII file with an extra unknown property and double precision
    txt = str(tmp_path / "t.ply")
    with open(txt, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\n")
        props = ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + ["opacity", "scale_0", "scale_1",
                 "rot_0", "rot_1", "rot_2", "rot_3", "confidence"]
        f.write("".join(f"property double {p}\n" for p in props) + "end_header\n")
        for r in range(2):
            f.write(" ".join(str(float(r * 100 + i)) for i in range(len(props))) + "\n")
    sc2 = ply_io.load_ply(txt)
    assert sc2.xyz.tolist() == [[0.0, 1.0, 2.0], [100.0, 101.0, 102.0]]
    assert sc2.features_rest.shape == (2, 15, 3) and float(sc2.features_rest[0, 1, 0]) == 7.0 and float(sc2.features_rest[0, 0, 1]) == 21.0
    with pytest.raises(ValueError):
        ply_io.load_ply(txt, max_sh_degree=2)


@pytest.mark.parametrize("tag", ["feat", "nofeat"])
def test_checkpoint_equals_what_the_reference_hands_to_plyfile(tmp_path, golden_dir, tag):
    """tests/golden/ply.npz (make_goldens.py): the reference's own ``save_ply`` / ``load_ply`` run with recording stand-ins
    for plyfile's two entry points - the vertex element's property names and float32 records exactly as the reference builds
    them (scene/gaussian_model.py:285-313, incl. the crop mask), and the tensors its ``load_ply`` (:364-417) builds back.
    Our writer's file = the PLY header for those names + those bytes, bit for bit; our reader returns the reference's
    tensors."""
    import os
    z = np.load(os.path.join(golden_dir, "ply.npz"))
    t = lambda k: torch.tensor(z[f"{tag}_in_{k}"])
    seg = t("seg_feature") if f"{tag}_in_seg_feature" in z else None
    crop = torch.tensor(z[f"{tag}_crop"]) if f"{tag}_crop" in z else None
    path = str(tmp_path / "point_cloud.ply")
    ply_io.save_ply(path, t("xyz"), t("f_dc"), t("f_rest"), t("opacity"), t("scaling"), t("rotation"), seg, crop_mask=crop)
    names = [str(n) for n in z[f"{tag}_names"]]
    body = z[f"{tag}_body"].tobytes()
    n = len(body) // (4 * len(names))
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n + "".join(f"property float {p}\n" for p in names) + "end_header\n"
    assert open(path, "rb").read() == header.encode() + body
    back = ply_io.load_ply(path, seg_feat_dim=(seg.shape[1] if seg is not None else None))
    for k, got in (("xyz", back.xyz), ("f_dc", back.features_dc), ("f_rest", back.features_rest), ("opacity", back.opacity_logit),
                   ("scaling", back.log_scale), ("rotation", back.rot)):
        np.testing.assert_array_equal(got.numpy(), z[f"{tag}_loaded_{k}"], err_msg=k)
    if seg is not None:
        np.testing.assert_array_equal(back.seg_feature.numpy(), z[f"{tag}_loaded_seg_feature"])
