"""CPU tests of the PLY ingest: the ASCII layout open3d writes for the reference, binary little endian,
colour scaling, pcgc_rescale, Simple_Render primitives."""
import numpy as np
import pytest

from pcrender import ply

O3D_ASCII = """ply
format ascii 1.0
comment Created by Open3D
element vertex 3
property double x
property double y
property double z
property double nx
property double ny
property double nz
property uchar red
property uchar green
property uchar blue
end_header
512 300.5 700 0 0 1 255 0 51
100 200 300 0 1 0 0 128 255
-1.5 2.25 1e3 1 0 0 10 20 30
"""


def test_reads_open3d_ascii(tmp_path):
    p = tmp_path / "pcd_0.ply"
    p.write_text(O3D_ASCII)
    d = ply.read_ply(str(p))
    np.testing.assert_array_equal(d["points"], [[512, 300.5, 700], [100, 200, 300], [-1.5, 2.25, 1000]])
    np.testing.assert_allclose(d["colors"], np.array([[255, 0, 51], [0, 128, 255], [10, 20, 30]]) / 255.0)
    np.testing.assert_array_equal(d["normals"], [[0, 0, 1], [0, 1, 0], [1, 0, 0]])
    np.testing.assert_allclose(ply.pcgc_rescale(d["points"][:1]), [[0.0, (300.5 - 512) / 256, (700 - 512) / 256]])


def test_round_trip_and_binary(tmp_path):
    rng = np.random.default_rng(0)
    pts, col = rng.uniform(0, 1023, (500, 3)), rng.integers(0, 256, (500, 3)) / 255.0
    a = tmp_path / "a.ply"
    ply.write_ply_ascii(str(a), pts, colors=col)
    d = ply.read_ply(str(a))
    np.testing.assert_allclose(d["points"], pts, rtol=1e-9)
    np.testing.assert_allclose(d["colors"], col, atol=1e-12)
    assert d["normals"] is None
    # binary little endian with float xyz + uchar colours (what many voxelised test sequences ship as)
    rec = np.zeros(500, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    rec["red"], rec["green"], rec["blue"] = (col * 255).round().astype(np.uint8).T
    b = tmp_path / "b.ply"
    hdr = ("ply\nformat binary_little_endian 1.0\nelement vertex 500\nproperty float x\nproperty float y\nproperty float z\n"
           "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
    b.write_bytes(hdr.encode() + rec.tobytes())
    e = ply.read_ply(str(b))
    np.testing.assert_allclose(e["points"], pts.astype(np.float32))
    np.testing.assert_allclose(e["colors"], col, atol=1e-12)


def test_errors(tmp_path):
    p = tmp_path / "x.ply"
    p.write_text("not a ply\n")
    with pytest.raises(ValueError):
        ply.read_ply(str(p))
    p.write_text("ply\nformat ascii 1.0\nelement vertex 2\nproperty double x\nproperty double y\nproperty double z\nend_header\n1 2 3\n")
    with pytest.raises(ValueError):
        ply.read_ply(str(p))


def test_simple_render_primitives():
    pts, col = np.zeros((4, 3)), np.full((4, 3), 0.75)
    g = ply.simple_render_primitives(pts, col, sigma=0.8, scale_factor=256.0, voxelized=True)
    assert g["shs"].shape == (4, 13, 3) and np.allclose(g["shs"][:, 0], 0.25 / 0.28209479177387814) and not g["shs"][:, 1:].any()
    assert np.allclose(g["scales"], 0.8 / 256.0) and (g["rotations"] == [1, 0, 0, 0]).all() and (g["opacities"] == 1).all()
