"""CPU tests of the synthetic workload generator: determinism, shapes, statistics."""
import numpy as np

from pcrender import synth


def test_clouds_are_deterministic_and_shaped():
    a = synth.make_cloud("synth-THuman-256", seed=0, P=5000)
    b = synth.make_cloud("synth-THuman-256", seed=0, P=5000)
    assert a["means3D"].tobytes() == b["means3D"].tobytes() and a["means3D"].shape == (5000, 3)
    # voxelised: coordinates sit on the 1/256 grid and are distinct (pcgc_rescale convention)
    q = a["means3D"].astype(np.float64) * 256
    assert np.abs(q - np.round(q)).max() < 1e-6 and np.unique(np.round(q), axis=0).shape[0] == 5000
    c = synth.make_cloud("synth-THuman-800K", seed=0, P=4000)
    assert c["scale_factor"] == 448.0 and c["means3D"].dtype == np.float32
    assert -1.3 < c["means3D"][:, 1].min() < -1.0 and 0.95 < c["means3D"][:, 1].max() < 1.1   # ~2.2 units tall, y up
    assert 0 <= c["rgb"].min() and c["rgb"].max() <= 1


def test_gaussian_parameters_follow_the_reference_conventions():
    c = synth.make_cloud("synth-THuman-800K", seed=0, P=3000)
    g = synth.make_gaussians(c, profile="training", seed=1)
    assert g["shs"].shape == (3000, 13, 3) and g["sh_degree"] == 1       # M=13 rows for degree 1 (quirk Q6)
    assert not g["shs"][:, 4:].any() and g["shs"][:, 1:4].any()
    np.testing.assert_allclose(g["shs"][:, 0], (c["rgb"] - 0.5) / synth.SH_C0, rtol=1e-5, atol=1e-6)
    radius = np.sqrt(3) / 448.0 * 6
    assert abs(np.median(g["scales"]) / radius - 1) < 0.05
    assert g["opacities"].shape == (3000, 1) and 0.2 <= g["opacities"].min() and g["opacities"].max() <= 1.0
    assert abs(np.linalg.norm(g["rotations"], axis=1).mean() - 1) < 0.05 and not np.allclose(np.linalg.norm(g["rotations"], axis=1), 1)
    gi = synth.make_gaussians(c, profile="inference", seed=1)
    assert (gi["opacities"] == 1).all() and not gi["shs"][:, 1:].any()


def test_batched_view_settings_are_bit_identical_to_per_view():
    import torch
    from pcrender import camera
    Hs = camera.circle_path(12, 0, 3, [90, 0])
    b = camera.raster_settings_batch(Hs, 960, 540, 45.0, 2)
    assert (b["image_height"], b["image_width"]) == (1080, 1920)
    for j in range(12):
        a = camera.raster_settings_arrays(Hs[j], 960, 540, 45.0, 2)
        assert a["tanfovx"] == b["tanfovx"] and a["tanfovy"] == b["tanfovy"]
        for k in ("viewmatrix", "projmatrix", "campos"):
            assert torch.equal(a[k], b[k][j]), (k, j)


def test_normals_per_view_match_the_reference_loop():
    """raster_passes.normals_per_view against the per-view loop of simple_raw_render.py:264-268 written out with the reference's
    tensor shapes (camera origin [1,1,3], hence a [1,N,1] sign tensor whose `[0]` strips the batch axis: per-POINT signs; the
    trace of the reference's own loop is in tests/test_cpu_call_trace.py)."""
    import torch
    from pcrender import camera, raster_passes as rp
    g = torch.Generator().manual_seed(5)
    Hs = camera.circle_path(12, 0, 3, [90, 0])
    for trial in range(20):
        means = torch.randn(50, 3, generator=g)
        normals = torch.nn.functional.normalize(torch.randn(50, 3, generator=g), dim=-1)
        if trial == 0:
            normals[0] = 0.0                     # the dot product of the first point is exactly 0 in every view
        cams = Hs[:, :3, 3]
        nv = rp.normals_per_view(means, normals, cams)
        colors = normals
        for j in range(12):
            camera_orig = Hs[None, j:j + 1, :3, 3]                               # Camera.get_camera_origin_w() of a [1,1,4,4] chunk
            sgn = (torch.sum((means - camera_orig) * colors, -1, keepdim=True) > 0).float() * 2 - 1
            colors = colors * (-1) * sgn[0]
            assert torch.equal(colors, nv[j]), (trial, j)
        # every normal faces its view's camera afterwards (or is perpendicular to the viewing ray)
        assert bool((torch.sum((means[None] - cams[:, None]) * nv, -1) <= 0).all())
