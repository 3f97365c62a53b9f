"""GPU tests at BASELINE.json's full sizes (the CPU oracle would need minutes here, so parity is checked
(a) bit-for-bit against the reference build, which runs these sizes in milliseconds on the GPU, and
(b) through size-independent properties: sortedness and stability of every tile list, ranges partitioning [0,R),
    conservation of pair counts, affine dependence on the background, linearity in precomputed colours,
    determinism, and first-order consistency of the gradient with a directional finite difference."""
import numpy as np
import pytest
import torch

import util
from util import run_product

pytestmark = pytest.mark.gpu

CONFIGS = {
    # BASELINE.json configs[1]: THuman-256 (200K voxelised), 1080p, inference profile
    "thuman256_1080p": dict(workload="synth-THuman-256", W=1920, H=1080, profile="inference", view=0),
    # configs[2]: THuman-800K, 1080p, training profile (the headline workload)
    "thuman800k_1080p": dict(workload="synth-THuman-800K", W=1920, H=1080, profile="training", view=3),
    # configs[4]: 2M-point sampled mesh, 4K
    "mesh2m_4k": dict(workload="synth-mesh-2M", W=3840, H=2160, profile="training", view=5),
}


def _scene(cfg, voxel_exact=False):
    from pcrender import camera, synth
    cloud = synth.make_cloud(cfg["workload"], seed=0)
    g = synth.make_gaussians(cloud, profile=cfg["profile"], seed=1)
    if voxel_exact:
        g["means3D"] = cloud["means3D"].copy()
    v = camera.circle_views(12, fov_deg=45.0, width_px=cfg["W"], height_px=cfg["H"])[cfg["view"]]
    return util.scene_from(g, v, cfg["W"], cfg["H"], bg=(1, 1, 1))


def _ref():
    return util.reference_build("strict")


@pytest.mark.parametrize("name", ["thuman256_1080p", "thuman800k_1080p", "mesh2m_4k"])
def test_fullsize_bit_exact_vs_reference_build(name, gpu_device):
    ref = _ref()
    s = _scene(CONFIGS[name], voxel_exact=(name == "thuman256_1080p"))   # exact voxel centres: depth keys tie massively
    r = ref.forward(s)
    p, _ = run_product(s, gpu_device)
    assert p["R"] == r["R"] and p["R"] > 5_000_000
    np.testing.assert_array_equal(p["radii"], r["radii"])
    np.testing.assert_array_equal(p["tiles_touched"], r["tiles_touched"])
    np.testing.assert_array_equal(p["vals"], r["vals"])
    np.testing.assert_array_equal(p["keys"], r["keys"])
    np.testing.assert_array_equal(p["ranges"], r["ranges"])
    np.testing.assert_array_equal(p["n_contrib"], r["n_contrib"])
    assert p["final_T"].tobytes() == r["final_T"].tobytes()
    assert p["out_color"].tobytes() == r["out_color"].tobytes()
    if name == "thuman256_1080p":
        k = r["keys"]
        assert (k[1:] == k[:-1]).mean() > 0.5, "expected mostly tied keys on the voxelised axis-aligned view"


def test_fullsize_backward_vs_reference_build(gpu_device):
    ref = _ref()
    s = _scene(CONFIGS["thuman800k_1080p"])
    dL = util.seeded_dL(s)
    _, gr = ref.forward_backward(s, dL)
    _, gp = run_product(s, gpu_device, dL_dpix=dL)
    util.check_grads(gp, gr, "800K/1080p vs reference build")


def test_fullsize_structural_properties(gpu_device):
    s = _scene(CONFIGS["thuman800k_1080p"])
    p, _ = run_product(s, gpu_device)
    R, keys, vals, ranges = p["R"], p["keys"], p["vals"], p["ranges"]
    assert R == int(p["tiles_touched"].sum()) == keys.size == vals.size
    # globally sorted by (tile, depth bits); ties broken by ascending Gaussian id (stability)
    assert (keys[1:] >= keys[:-1]).all()
    tie = keys[1:] == keys[:-1]
    assert (vals[1:][tie] > vals[:-1][tie]).all()
    # every pair's key is (tile of the range it sits in) | depth bits of its Gaussian
    depth_bits = p["depths"].view(np.uint32)[vals].astype(np.uint64)
    assert ((keys & np.uint64(0xFFFFFFFF)) == depth_bits).all()
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    nz = np.nonzero(ranges[:, 1] > ranges[:, 0])[0]
    assert ranges[nz[0], 0] == 0 and ranges[nz[-1], 1] == R
    assert (ranges[nz[1:], 0] == ranges[nz[:-1], 1]).all()
    assert (np.repeat(nz, (ranges[nz, 1] - ranges[nz, 0]).astype(np.int64)) == tiles).all()
    # per-pixel bookkeeping
    gx = (s.W + 15) // 16
    H, W = s.H, s.W
    tile_of_pixel = (np.arange(H)[:, None] // 16) * gx + (np.arange(W)[None, :] // 16)
    lens = (ranges[:, 1] - ranges[:, 0])[tile_of_pixel]
    assert (p["n_contrib"] <= lens).all()
    assert p["final_T"].min() >= 1e-4 * (1 - 0.99) and p["final_T"].max() <= 1.0
    assert ((p["n_contrib"] == 0) <= (p["final_T"] == 1.0)).all()      # untouched pixels keep T = 1


def test_fullsize_background_affinity_and_determinism(gpu_device):
    s = _scene(CONFIGS["thuman800k_1080p"])
    a, _ = run_product(s, gpu_device)
    b, _ = run_product(s, gpu_device)
    assert a["out_color"].tobytes() == b["out_color"].tobytes() and np.array_equal(a["vals"], b["vals"])
    s0 = _scene(CONFIGS["thuman800k_1080p"])
    s0.bg[:] = 0
    z, _ = run_product(s0, gpu_device)
    assert np.array_equal(z["final_T"], a["final_T"]) and np.array_equal(z["n_contrib"], a["n_contrib"])
    np.testing.assert_allclose(a["out_color"], z["out_color"] + a["final_T"][None] * s.bg[:, None, None], atol=2e-7, rtol=0)


def test_fullsize_linearity_in_precomputed_colours(gpu_device):
    """With colors_precomp the image is linear in the colours (same lists, same alphas, same T)."""
    cfg = CONFIGS["thuman256_1080p"]
    base = _scene(cfg)
    rng = np.random.default_rng(11)
    P = base.P
    c1 = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    c2 = rng.uniform(0, 1, (P, 3)).astype(np.float32)

    def render(c):
        from oracle.oracle import Scene
        s = Scene(W=base.W, H=base.H, tanfovx=base.tanfovx, tanfovy=base.tanfovy, bg=np.zeros(3, np.float32),
                  means3D=base.means3D, opacities=base.opacities, viewmatrix=base.viewmatrix, projmatrix=base.projmatrix,
                  campos=base.campos, colors_precomp=c, scales=base.scales, rotations=base.rotations)
        return run_product(s, gpu_device)[0]["out_color"].astype(np.float64)

    np.testing.assert_allclose(render(c1 + c2), render(c1) + render(c2), atol=5e-6, rtol=0)


def test_fullsize_gradient_matches_directional_finite_difference(gpu_device):
    """loss(theta + eps d) - loss(theta - eps d) ~ 2 eps <grad, d> for a random direction in opacity + SH space (both
    enter the image smoothly; positions/scales move tile membership and are covered at small size by the oracle)."""
    from oracle.oracle import Scene
    s = _scene(CONFIGS["thuman800k_1080p"])
    dL = util.seeded_dL(s)
    _, g = run_product(s, gpu_device, dL_dpix=dL)
    rng = np.random.default_rng(21)
    d_op = rng.standard_normal(s.opacities.shape).astype(np.float32)
    d_sh = np.zeros_like(s.shs)
    d_sh[:, :4] = rng.standard_normal((s.P, 4, 3)).astype(np.float32)
    eps = 1e-3

    def loss(sign):
        s2 = Scene(W=s.W, H=s.H, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=s.bg, means3D=s.means3D,
                   opacities=s.opacities + sign * eps * d_op, viewmatrix=s.viewmatrix, projmatrix=s.projmatrix,
                   campos=s.campos, shs=s.shs + sign * eps * d_sh, scales=s.scales, rotations=s.rotations, sh_degree=1)
        return float((run_product(s2, gpu_device)[0]["out_color"].astype(np.float64) * dL).sum())

    fd = (loss(+1) - loss(-1)) / (2 * eps)
    t_op = g["dL_dopacity"].astype(np.float64).ravel() * d_op.ravel()
    t_sh = g["dL_dsh"].astype(np.float64) * d_sh
    an = float(t_op.sum() + t_sh.sum())
    # the random direction makes the directional derivative a heavily cancelling sum; the loss also has kinks (alpha
    # cut at 1/255, 0.99 clamp, T < 1e-4 stop, SH clamp at 0), so agreement is asked relative to the sum of |terms|
    scale = float(np.abs(t_op).sum() + np.abs(t_sh).sum())
    assert abs(fd - an) <= 2e-3 * scale, (fd, an, scale)


def test_more_than_65536_tiles_uses_32_bit_tile_keys(gpu_device):
    """4096 x 4112 pixels = 256 x 257 = 65 792 tiles: tile ids no longer fit the 16-bit keys of the usual tile sort, and
    the tile-id part of the reference's key needs 17 bits (three 8-bit-or-less passes here)."""
    from pcrender import synth
    ref = _ref()
    W, H = 4096, 4112
    g = synth.random_scene(3000, W, H, seed=31, sh_degree=1, spread=1.6, scale=0.02)
    s = util.scene_from(g, util.identity_camera(W, H), W, H, bg=(0.1, 0.2, 0.3))
    r = ref.forward(s)
    p, _ = run_product(s, gpu_device)
    assert p["R"] == r["R"] and p["R"] > 10_000
    assert int(r["keys"].max() >> np.uint64(32)) >= 65536, "scene does not reach a tile id beyond 16 bits"
    for k in ("radii", "tiles_touched", "vals", "keys", "ranges", "n_contrib"):
        np.testing.assert_array_equal(p[k], r[k], err_msg=k)
    assert p["out_color"].tobytes() == r["out_color"].tobytes()
    dL = util.seeded_dL(s)
    _, gr = ref.forward_backward(s, dL)
    _, gp = run_product(s, gpu_device, dL_dpix=dL)
    util.check_grads(gp, gr, ">65536 tiles vs reference build")


def test_six_million_points_forward_backward_vs_reference_build(gpu_device):
    """Beyond BASELINE's largest cloud: 6 M Gaussians at 1080p (three times configs[4]'s point count: per-Gaussian indices times the
    13 SH rows pass 2^27, the depth sort runs 1 465 whole-size blocks, ~90 M pairs go through the tile sort of ONE view), forward and
    backward against the reference build: integers and image bit-identical, gradients inside the bars."""
    from pcrender import camera, synth
    ref = _ref()
    cloud = synth.make_cloud("synth-mesh-2M", seed=3, P=6_000_000)
    g = synth.make_gaussians(cloud, profile="training", seed=4)
    W, H = 1920, 1080
    v = camera.circle_views(12, fov_deg=45.0, width_px=W, height_px=H)[2]
    s = util.scene_from(g, v, W, H, bg=(1, 1, 1))
    dL = util.seeded_dL(s)
    r, gr = ref.forward_backward(s, dL)
    p, gp = run_product(s, gpu_device, dL_dpix=dL, light=True, reference_lists=False)
    assert p["R"] == r["R"] and p["R"] > 60_000_000
    np.testing.assert_array_equal(p["radii"], r["radii"])
    assert p["out_color"].tobytes() == r["out_color"].tobytes()
    util.check_grads(gp, gr, "6M points / 1080p vs reference build")
