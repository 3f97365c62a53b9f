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
