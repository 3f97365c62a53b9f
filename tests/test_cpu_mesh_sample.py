"""Mesh -> point-cloud sampling: quantisation / dedupe arithmetic of the reference
(structures.py:3876-3888) and the statistical properties of the sampler."""
import numpy as np
import pytest

from pcrender import mesh_sample as ms


def _cube():
    v = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float64) - 0.5
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    f = np.array([t for q in quads for t in ((q[0], q[1], q[2]), (q[0], q[2], q[3]))], np.int64)
    return v, f


def test_quantize_dedupe_is_the_reference_arithmetic():
    rng = np.random.default_rng(0)
    p = rng.uniform(-1, 1, (5000, 3))
    a = rng.uniform(0, 1, (5000, 3))
    q, a_q = ms.quantize_dedupe(p, a)
    # the reference's three lines, verbatim semantics
    ref = np.round(p * 448)
    ref += 512
    _, idx = np.unique(ref, axis=0, return_index=True)
    np.testing.assert_array_equal(q, ref[idx])
    np.testing.assert_array_equal(a_q, a[idx])
    # lexicographic voxel order, no duplicates, first sample wins
    assert (np.diff(q.view([("", q.dtype)] * 3).ravel().argsort(kind="stable")) == 1).all()
    assert len({tuple(r) for r in q}) == len(q)
    dup = np.array([[0.1, 0.2, 0.3], [0.1004, 0.2001, 0.2996], [0.9, 0.9, 0.9]])
    qd, tag = ms.quantize_dedupe(dup, np.array([10, 20, 30]))
    assert len(qd) == 2 and set(tag.tolist()) == {10, 30}
    np.testing.assert_allclose(ms.to_gaussian_means(qd), np.round(dup[[0, 2]] * 448) / 448, atol=1e-6)


def test_uniform_sampling_is_area_weighted_and_on_the_surface():
    v, f = _cube()
    v = v * np.array([2.0, 1.0, 0.5])        # faces of different areas
    s = ms.sample_uniform(v, f, 60000, seed=1)
    x = s["xyz"]
    # every sample lies on one of the six faces
    on = np.isclose(np.abs(x), np.array([1.0, 0.5, 0.25]), atol=1e-9)
    assert on.any(axis=1).all()
    # share of samples per triangle ~ triangle area share
    area = ms.triangle_areas(v, f)
    share = np.bincount(s["face"], minlength=len(f)) / len(x)
    np.testing.assert_allclose(share, area / area.sum(), atol=0.01)
    # uniform inside a triangle: barycentre of the samples of one triangle ~ its centroid
    t = 0
    np.testing.assert_allclose(x[s["face"] == t].mean(0), v[f[t]].mean(0), atol=0.02)
    # normals are unit and perpendicular to their face on this flat-shaded shape (up to vertex-normal blending)
    assert np.allclose(np.linalg.norm(s["normal"], axis=1), 1.0, atol=1e-9)
    # determinism
    np.testing.assert_array_equal(ms.sample_uniform(v, f, 100, seed=5)["xyz"], ms.sample_uniform(v, f, 100, seed=5)["xyz"])


def test_obj_round_trip_and_point_cloud(tmp_path):
    v, f = _cube()
    col = (v + 0.5)
    path = tmp_path / "cube.obj"
    with open(path, "w") as fh:
        fh.write("# cube\n")
        for p, c in zip(v, col):
            fh.write("v %f %f %f %f %f %f\n" % (*p, *c))
        for q in range(0, len(f), 2):           # write quads: the reader must fan them back into the same triangles
            a, b, c = f[q]
            d = f[q + 1][2]
            fh.write("f %d %d %d %d\n" % (a + 1, b + 1, c + 1, d + 1))
    m = ms.read_obj(str(path))
    np.testing.assert_allclose(m["vertices"], v)
    np.testing.assert_array_equal(m["faces"], f)
    np.testing.assert_allclose(m["colors"], col, atol=1e-6)
    pc = ms.sample_point_cloud(m, 20000, method="uniform_quantized", seed=3)
    q = pc["xyz_w"]
    assert q.dtype == np.float32 and pc["rgb"].shape == q.shape and pc["normal_w"].shape == q.shape
    assert (q == np.round(q)).all() and q.min() >= 512 - 224 and q.max() <= 512 + 224   # cube of side 1 -> 448 voxels
    assert len(np.unique(q, axis=0)) == len(q) <= 20000
    # colour was interpolated from the vertices: on this cube it equals position + 0.5 (before quantisation error)
    np.testing.assert_allclose(pc["rgb"], (q - 512) / 448 + 0.5, atol=2.0 / 448)
    pu = ms.sample_point_cloud(m, 1000, method="uniform", seed=3)
    assert pu["xyz_w"].shape == (1000, 3)
    with pytest.raises(NotImplementedError):
        ms.sample_point_cloud(m, 10, method="poisson_disk")


# ------------------------------------------------------------------------------------------- textured meshes (OBJ + MTL + image)
def _write_textured_quad(d, flip_second=False, plain_first=False):
    """A unit quad in the z = 0 plane (two triangles, CCW seen from +z) with vt spanning the whole texture, a 4x3 texture whose
    texels are all different, plus a second quad with its own material / texture."""
    from PIL import Image
    tex = np.zeros((3, 4, 3), np.uint8)
    for y in range(3):
        for x in range(4):
            tex[y, x] = (40 * x + 10, 60 * y + 20, 200 - 30 * x - 20 * y)
    Image.fromarray(tex).save(str(d / "a.png"))
    tex2 = np.full((2, 2, 3), 255, np.uint8)
    tex2[0, 0] = (255, 0, 0)
    Image.fromarray(tex2).save(str(d / "b.png"))
    if plain_first:   # an untextured material BETWEEN the textured ones: the reference's numbering quirk (read_obj docstring)
        (d / "quad.mtl").write_text("newmtl matA\nKd 1 1 1\nmap_Kd a.png\n\nnewmtl plain\nKd 0.5 0.5 0.5\n\nnewmtl matB\nmap_Kd b.png\n")
    else:
        (d / "quad.mtl").write_text("newmtl matA\nKd 1 1 1\nmap_Kd a.png\n\nnewmtl matB\nmap_Kd b.png\n\nnewmtl plain\nKd 0.5 0.5 0.5\n")
    (d / "quad.obj").write_text(
        "mtllib quad.mtl\n"
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\n"
        "v 2 0 0\nv 3 0 0\nv 3 1 0\nv 2 1 0\n"
        "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
        "vn 0 0 1\n"
        "usemtl matA\nf 1/1/1 2/2/1 3/3/1 4/4/1\n"
        "usemtl matB\nf 5/1/1 6/2/1 7/3/1 8/4/1\n")
    return tex, tex2


def test_uv_lookup_matches_the_reference_uvmap():
    """golden vectors generated from /root/reference/plib/uv_mapping.py::UVMap (tests/golden/make_golden_py.py uvmap)"""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "py_uvmap.npz"))
    got = ms.uv_lookup(d["texture"], d["uv"])
    np.testing.assert_allclose(got, d["result"], rtol=0, atol=1e-12)


def test_material_ids_follow_open3d_numbering(tmp_path):
    """open3d keeps tinyobjloader's material index over ALL materials of the MTL in triangle_material_ids while mesh.textures
    holds only the materials with a map_Kd, and the reference pairs texture t with material id t (plib/render.py:158-173):
    with MTL order matA (textured), plain, matB (textured), matB's triangles (id 2) match no texture and come out black.
    material_numbering='textured' is the repaired numbering."""
    _write_textured_quad(tmp_path, plain_first=True)
    mesh = ms.read_obj(str(tmp_path / "quad.obj"))
    assert len(mesh["textures"]) == 2 and mesh["material_ids"].tolist() == [0, 0, 2, 2]
    s = ms.sample_uniform(mesh["vertices"], mesh["faces"], 4000, seed=3, triangle_uvs=mesh["triangle_uvs"],
                          material_ids=mesh["material_ids"], textures=mesh["textures"])
    right = s["xyz"][:, 0] > 1.5
    assert right.sum() > 1000 and np.all(s["rgb"][right] == 0.0) and s["rgb"][~right].max() > 0.1
    fixed = ms.read_obj(str(tmp_path / "quad.obj"), material_numbering="textured")
    assert fixed["material_ids"].tolist() == [0, 0, 1, 1]
    # a usemtl the MTL does not define: tinyobjloader's -1
    (tmp_path / "q2.obj").write_text((tmp_path / "quad.obj").read_text().replace("usemtl matB", "usemtl nowhere"))
    assert ms.read_obj(str(tmp_path / "q2.obj"))["material_ids"].tolist() == [0, 0, -1, -1]


def test_textured_obj_colours_come_from_the_texture(tmp_path):
    tex, tex2 = _write_textured_quad(tmp_path)
    mesh = ms.read_obj(str(tmp_path / "quad.obj"))
    assert mesh["triangle_uvs"].shape == (4, 3, 2) and len(mesh["textures"]) == 2
    assert mesh["material_ids"].tolist() == [0, 0, 1, 1]          # MTL order matA, matB, plain: material id == texture index
    # the image is stored flipped vertically, as open3d's OBJ reader does
    np.testing.assert_array_equal(mesh["textures"][0], tex[::-1].astype(np.float32) / np.float32(255))
    s = ms.sample_uniform(mesh["vertices"], mesh["faces"], 20000, seed=3, triangle_uvs=mesh["triangle_uvs"],
                          material_ids=mesh["material_ids"], textures=mesh["textures"])
    xyz, rgb = s["xyz"], s["rgb"]
    left = xyz[:, 0] < 1.5
    assert left.sum() > 5000 and (~left).sum() > 5000
    # quad A: uv == (x, y) of the sample, so the colour is the bilinear texture value at (u, v) with v = 1 at the image's TOP row
    # (OBJ convention); expected value from the unflipped image: row coordinate (1 - v) * h - 0.5
    u, v = xyz[left, 0], xyz[left, 1]
    h, w = tex.shape[:2]
    t = tex.astype(np.float64) / 255.0
    yy, xx = (1.0 - v) * h - 0.5, u * w - 0.5
    y0, x0 = np.floor(yy).astype(int), np.floor(xx).astype(int)
    fy, fx = (yy - y0)[:, None], (xx - x0)[:, None]
    g = lambda a, b: t[np.mod(a, h), np.mod(b, w)]  # noqa: E731  (wrap padding)
    want = (g(y0, x0) * (1 - fx) + g(y0, x0 + 1) * fx) * (1 - fy) + (g(y0 + 1, x0) * (1 - fx) + g(y0 + 1, x0 + 1) * fx) * fy
    np.testing.assert_allclose(rgb[left], want, atol=1e-6)
    # quad B uses the second texture only: red comes from its top-left texel, everything else is white
    rb = rgb[~left]
    assert rb.min() >= -1e-9 and rb.max() <= 1 + 1e-9 and np.allclose(rb[:, 0], 1.0, atol=1e-6)
    ub, vb = xyz[~left, 0] - 2.0, xyz[~left, 1]
    np.testing.assert_allclose(rb, ms.uv_lookup(mesh["textures"][1], np.stack([ub, vb], 1)), atol=1e-12)   # texture 1, not texture 0
    centre = (np.abs(ub - 0.25) < 0.02) & (np.abs(vb - 0.75) < 0.02)    # around the centre of the top-left texel: (almost) pure red
    assert centre.sum() > 5 and rb[centre][:, 1:].max() < 0.2
    # normals: (0,0,1) interpolated, then turned against the ray direction (1,1,1) -> (0,0,-1)   (structures.py:3776-3780)
    np.testing.assert_allclose(s["normal"], np.tile([0.0, 0.0, -1.0], (xyz.shape[0], 1)), atol=1e-12)
    # the whole pipeline, quantised: attributes follow the first sample of every voxel
    pc = ms.sample_point_cloud(mesh, 5000, method="uniform_quantized", seed=3)
    assert pc["rgb"].shape == pc["xyz_w"].shape and pc["rgb"].dtype == np.float32
    pc0 = ms.sample_point_cloud(dict(vertices=mesh["vertices"], faces=mesh["faces"]), 100, method="uniform", seed=1)
    assert np.array_equal(pc0["rgb"], np.ones_like(pc0["xyz_w"]))        # no texture, no vertex colours: ones
