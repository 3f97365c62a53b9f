"""Shared test helpers: seeded scenes (numpy), and runners that put the CPU oracle, the reference build and
the HIP product behind one dict-of-numpy-arrays interface."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gaussian-pcloud-render_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from oracle.oracle import Scene  # noqa: E402
from pcrender import camera as cam  # noqa: E402
from pcrender import synth  # noqa: E402


# counts the comparisons made against the reference build in this process (smoke() and the fuzz tally print it)
REF_USES = {"strict": 0, "fast": 0}


def reference_build(variant="strict"):
    """The reference's own kernels built for gfx950 (oracle/_ref/libgsr_ref_<variant>.so, made by oracle/build_ref.sh where
    /root/reference exists; git-ignored, it rides to the GPU box with the push).  The -m gpu tests that compare against it are
    most of the parity evidence, so a library that did not travel FAILS them -- a green run must not depend on luck.  Only
    GSR_ALLOW_NO_REF=1 (a GPU machine that never had the reference mounted) turns the failure into a skip."""
    import pytest
    from oracle.oracle import REF_SO, Reference
    if not Reference.available(variant):
        msg = ("%s is missing: oracle/_ref/*.so did not travel (build it with oracle/build_ref.sh where /root/reference is "
               "mounted; GSR_ALLOW_NO_REF=1 skips instead)" % REF_SO[variant])
        if os.environ.get("GSR_ALLOW_NO_REF", "0") == "1" and os.environ.get("GSR_REQUIRE_REF", "0") != "1":
            pytest.skip(msg)
        pytest.fail(msg)
    REF_USES[variant] += 1
    return Reference(variant)


def _view_arrays(H_c2w, W, H, fov_deg):
    import torch
    s = cam.raster_settings_arrays(torch.as_tensor(H_c2w, dtype=torch.float32), W, H, fov_deg, 1)
    return s


def identity_camera(W, H, fov_deg=60.0):
    """Camera at the origin looking down +z, settings built exactly like the reference caller would."""
    return _view_arrays(np.eye(4, dtype=np.float32), W, H, fov_deg)


def scene_from(g, view, W, H, bg=(0.0, 0.0, 0.0), mode="sh", scale_modifier=1.0, sh_degree=None, use_cov3d=False):
    """g: dict from synth.random_scene / make_gaussians; view: dict from camera.raster_settings_arrays."""
    kw = dict(W=W, H=H, tanfovx=view["tanfovx"], tanfovy=view["tanfovy"], bg=np.asarray(bg, np.float32),
              means3D=g["means3D"], opacities=g["opacities"], viewmatrix=np.asarray(view["viewmatrix"]),
              projmatrix=np.asarray(view["projmatrix"]), campos=np.asarray(view["campos"]),
              scale_modifier=scale_modifier)
    if mode == "sh":
        kw.update(shs=g["shs"], sh_degree=g["sh_degree"] if sh_degree is None else sh_degree)
    else:
        kw.update(colors_precomp=g["colors_precomp"])
    if use_cov3d:
        kw.update(cov3D_precomp=cov3d_from(g["scales"], g["rotations"], scale_modifier))
    else:
        kw.update(scales=g["scales"], rotations=g["rotations"])
    return Scene(**kw)


def cov3d_from(scales, rotations, mod=1.0):
    """fp64 numpy 3D covariance (R S^2 R^T upper triangle) used only to build cov3D_precomp inputs."""
    s = scales.astype(np.float64) * mod
    q = rotations.astype(np.float64)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
        np.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
        np.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], -2)
    S2 = s[:, None, :] ** 2
    Sig = (R * S2) @ np.transpose(R, (0, 2, 1))
    return np.stack([Sig[:, 0, 0], Sig[:, 0, 1], Sig[:, 0, 2], Sig[:, 1, 1], Sig[:, 1, 2], Sig[:, 2, 2]], -1).astype(np.float32)


# ---------------------------------------------------------------------------------------------- scenes
def build_scene(name):
    """Named, seeded scenes small enough for the CPU oracle to finish in seconds."""
    if name == "random_aniso":           # non-multiple-of-16 image, SH degree 1 in a 13-row tensor (reference M=13)
        W, H = 80, 56
        g = synth.random_scene(3000, W, H, seed=0, sh_degree=1, sh_rows=13)
        return scene_from(g, identity_camera(W, H), W, H, bg=(1, 1, 1))
    if name.startswith("sh_deg"):        # sh_deg0..3 with the full 16 rows
        D = int(name[-1])
        W, H = 96, 64
        g = synth.random_scene(2000, W, H, seed=10 + D, sh_degree=D, sh_rows=16)
        return scene_from(g, identity_camera(W, H), W, H, bg=(0.2, 0.4, 0.6))
    if name == "colors_precomp":
        W, H = 64, 64
        g = synth.random_scene(1500, W, H, seed=3, sh_degree=0)
        return scene_from(g, identity_camera(W, H), W, H, bg=(0, 0, 0), mode="colors")
    if name == "cov3d_precomp":
        W, H = 64, 48
        g = synth.random_scene(1500, W, H, seed=4, sh_degree=1)
        return scene_from(g, identity_camera(W, H), W, H, bg=(0.5, 0.5, 0.5), use_cov3d=True, scale_modifier=1.0)
    if name == "scale_modifier":
        W, H = 64, 64
        g = synth.random_scene(1000, W, H, seed=5, sh_degree=1)
        return scene_from(g, identity_camera(W, H), W, H, bg=(1, 0, 1), scale_modifier=1.7)
    if name == "culled_mix":             # behind the near plane, off-screen, huge, tiny
        W, H = 100, 70
        g = synth.random_scene(4000, W, H, seed=6, sh_degree=1, spread=6.0)
        g["means3D"][::5, 2] = np.linspace(-3, 0.25, g["means3D"][::5].shape[0])   # around the 0.2 near plane
        g["means3D"][7, 2] = 0.2                                                  # exactly on the plane: culled (<=)
        g["scales"][::11] *= 30.0
        g["scales"][3::17] *= 1e-3
        return scene_from(g, identity_camera(W, H), W, H, bg=(0.1, 0.2, 0.3))
    if name == "all_culled":             # R == 0: image is pure background
        W, H = 48, 32
        g = synth.random_scene(300, W, H, seed=7, sh_degree=0)
        g["means3D"][:, 2] = -1.0
        return scene_from(g, identity_camera(W, H), W, H, bg=(0.3, 0.6, 0.9))
    if name == "voxel_ties":             # voxel grid seen by an axis-aligned camera: almost all depth keys tie
        W, H = 128, 128
        rng = np.random.default_rng(8)
        n = 24
        ii, jj, kk = np.meshgrid(np.arange(n), np.arange(n), np.arange(4), indexing="ij")
        means = np.stack([(ii - n / 2) / 16.0, (jj - n / 2) / 16.0, 2.0 + kk / 16.0], -1).reshape(-1, 3)
        perm = rng.permutation(means.shape[0])   # ids unrelated to position so tie order is observable
        means = means[perm].astype(np.float32)
        P = means.shape[0]
        g = dict(means3D=means, scales=np.full((P, 3), 0.03, np.float32),
                 rotations=np.tile(np.array([1, 0, 0, 0], np.float32), (P, 1)),
                 opacities=rng.uniform(0.3, 0.9, (P, 1)).astype(np.float32),
                 shs=(0.5 * rng.standard_normal((P, 4, 3))).astype(np.float32), sh_degree=1)
        return scene_from(g, identity_camera(W, H, 50.0), W, H, bg=(0, 0, 0))
    if name == "opaque_early_stop":      # opacity 1 everywhere: exercises the T < 1e-4 termination and alpha clamp 0.99
        W, H = 64, 64
        g = synth.random_scene(6000, W, H, seed=9, sh_degree=0, spread=0.6, scale=0.08)
        g["opacities"][:] = 1.0
        return scene_from(g, identity_camera(W, H), W, H, bg=(1, 1, 1))
    if name == "capsule_circle":         # the benchmark's synthetic body + the reference's circle camera, view 1
        W, H = 256, 256
        cloud = synth.make_cloud("synth-THuman-256", seed=0, P=20000)
        g = synth.make_gaussians(cloud, profile="training", seed=1)
        views = cam.circle_views(n_imgs=12, fov_deg=45.0, width_px=W, height_px=H)
        return scene_from(g, views[1], W, H, bg=(1, 1, 1))
    if name == "capsule_axis_view":      # view 0 of the circle is axis aligned; voxelised cloud -> heavy depth ties
        W, H = 192, 160
        cloud = synth.make_cloud("synth-THuman-256", seed=0, P=15000)
        g = synth.make_gaussians(cloud, profile="inference", seed=1)
        g["means3D"] = cloud["means3D"].copy()     # keep exact voxel centres (no learned offsets) so depths tie
        views = cam.circle_views(n_imgs=12, fov_deg=45.0, width_px=W, height_px=H)
        return scene_from(g, views[0], W, H, bg=(1, 1, 1))
    if name == "big_splats":             # large footprints: ~40 tiles per Gaussian, multi-round tile lists
        W, H = 160, 112
        g = synth.random_scene(2500, W, H, seed=12, sh_degree=1, spread=1.5, scale=0.25)
        g["opacities"] *= 0.35
        return scene_from(g, identity_camera(W, H), W, H, bg=(0.9, 0.8, 0.7))
    if name == "deep_stack":             # one 20 000-entry list that no pixel ever terminates: more than BWD_MAX_CHUNKS slices
        W, H = 24, 16                    # (two tiles, one of them partial)
        rng = np.random.default_rng(13)
        P = 20000
        means = np.stack([rng.uniform(-0.25, 0.25, P), rng.uniform(-0.2, 0.2, P), rng.uniform(1.0, 3.0, P)], 1).astype(np.float32)
        g = dict(means3D=means, scales=np.exp(rng.normal(np.log(0.08), 0.3, (P, 3))).astype(np.float32),
                 rotations=np.tile(np.array([1, 0, 0, 0], np.float32), (P, 1)),
                 opacities=rng.uniform(0.004, 0.0075, (P, 1)).astype(np.float32),   # alpha hovers around the 1/255 cut
                 shs=(0.6 * rng.standard_normal((P, 4, 3))).astype(np.float32), sh_degree=1)
        return scene_from(g, identity_camera(W, H), W, H, bg=(0.2, 0.3, 0.4))
    if name.startswith("depth_span_"):   # depth keys that differ in 8 / 16 / 24 / 32 bits: the depth sort runs 1 / 2 / 3 / 4 passes
        npass = int(name[-1])
        W, H = 96, 64
        rng = np.random.default_rng(40 + npass)
        g = synth.random_scene(1800, W, H, seed=30 + npass, sh_degree=1)
        P = g["means3D"].shape[0]
        if npass == 4:
            z = np.exp(rng.uniform(np.log(0.25), np.log(60.0), P)).astype(np.float32)       # eight binades
        else:
            # codes base .. base + span - 1 straddling the binade boundary at 2.0 (identity camera: view-space z = z);
            # base is a multiple of 256, so the span is what the sort sees
            span = {1: 200, 2: 60000, 3: 6000000}[npass]
            base = (np.float32(2.0).view(np.uint32) - np.uint32(span // 2)) & np.uint32(0xFFFFFF00)
            z = (base + rng.integers(0, span, P).astype(np.uint32)).view(np.float32)
            z[:2] = np.array([base, base + np.uint32(span - 1)], np.uint32).view(np.float32)  # both ends present
        g["means3D"][:, 0] *= z / g["means3D"][:, 2]      # keep the screen position
        g["means3D"][:, 1] *= z / g["means3D"][:, 2]
        g["means3D"][:, 2] = z
        g["means3D"][5::9, 2] = -1.0                       # culled Gaussians (key 0xFFFFFFFF) do not count
        return scene_from(g, identity_camera(W, H), W, H, bg=(0.3, 0.3, 0.3))
    if name == "one_gaussian":
        W, H = 33, 17
        g = dict(means3D=np.array([[0.05, -0.02, 1.5]], np.float32), scales=np.array([[0.2, 0.05, 0.1]], np.float32),
                 rotations=np.array([[0.9, 0.1, 0.3, -0.2]], np.float32), opacities=np.array([[0.8]], np.float32),
                 shs=np.array([[[0.5, -0.2, 0.1]] * 4], np.float32), sh_degree=1)
        return scene_from(g, identity_camera(W, H), W, H, bg=(0.25, 0.5, 0.75))
    raise KeyError(name)


SCENES = ["random_aniso", "sh_deg0", "sh_deg1", "sh_deg2", "sh_deg3", "colors_precomp", "cov3d_precomp",
          "scale_modifier", "culled_mix", "all_culled", "voxel_ties", "opaque_early_stop", "capsule_circle",
          "capsule_axis_view", "big_splats", "deep_stack", "one_gaussian", "depth_span_1", "depth_span_2", "depth_span_3",
          "depth_span_4"]


def seeded_dL(scene, seed=123):
    return np.random.default_rng(seed).uniform(-1, 1, (3, scene.H, scene.W)).astype(np.float32)


# ------------------------------------------------------------------------------------------ product runner
def run_product(scene, device, dL_dpix=None, debug=False, need_backward=None, light=False, reference_lists=True):
    """Run the HIP library through the package's native binding; returns (forward dict, grads dict or None) with the
    same keys/shapes as oracle.Oracle.forward.  light=True skips the copies of the private arena arrays (only R, the
    image and radii are returned): for full-size scenes where they are not compared.
    reference_lists=True (the default HERE, not in the product): the reference's full tile rectangles are emitted, so the
    private lists / ranges / n_contrib can be compared with the reference's element by element.  False = the product's default,
    footprint clipping (include/gsr.h): every API-visible result must be the same, the lists are sub-lists ("L" pairs instead
    of "R"); check_clipped_equivalent below holds the two modes against each other."""
    from diff_gaussian_rasterization import _native as N
    old = N.set_reference_lists(reference_lists)
    try:
        return _run_product(scene, device, dL_dpix, debug, need_backward, light)
    finally:
        N.set_reference_lists(old)


def _run_product(scene, device, dL_dpix, debug, need_backward, light):
    import torch
    from diff_gaussian_rasterization import _native as N

    def t(a):
        return torch.empty(0) if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(device)

    nb = (dL_dpix is not None) if need_backward is None else need_backward
    args = (t(scene.bg), t(scene.means3D), t(scene.colors_precomp), t(scene.opacities), t(scene.scales),
            t(scene.rotations), scene.scale_modifier, t(scene.cov3D_precomp), t(scene.viewmatrix.reshape(4, 4)),
            t(scene.projmatrix.reshape(4, 4)), scene.tanfovx, scene.tanfovy, scene.H, scene.W, t(scene.shs),
            scene.sh_degree, t(scene.campos), scene.prefiltered, debug)
    R, color, radii, geom, binning, img = N.rasterize_gaussians(*args, need_backward=nb)
    P, W, H = scene.P, scene.W, scene.H
    out = dict(P=P, W=W, H=H, R=R, out_color=color.cpu().numpy(), radii=radii.cpu().numpy())
    if P:
        lp = N.query("LIST_PAIRS", P, W, H, R, geom, binning, img).cpu().numpy()
        out["L"] = int(lp[0])                       # pairs in the library's lists (<= R with footprint clipping)
        assert int(lp[1]) == R, "the device's reference pair count differs from the reported num_rendered"
    if P and not light:
        def q(name):
            return N.query(name, P, W, H, out["L"], geom, binning, img).cpu().numpy()
        out.update(
            depths=q("DEPTHS"), means2D=q("MEANS2D"), conic_opacity=q("CONIC_OPACITY"), rgb=q("RGB"),
            tiles_touched=q("TILES_TOUCHED").view(np.uint32), vals=q("POINT_LIST").view(np.uint32),
            keys=q("POINT_LIST_KEYS").view(np.uint64), ranges=q("RANGES").view(np.uint32),
            final_T=q("FINAL_T").reshape(H, W), n_contrib=q("N_CONTRIB").view(np.uint32).reshape(H, W),
        )
        out["depth_sort"] = q("DEPTH_SORT").view(np.uint32)   # key base, key bits compared, passes run
        if nb:
            out["clamped"] = q("CLAMPED")
            out["tile_need"] = q("TILE_NEED").view(np.uint32)   # list entries each tile's forward walked: what the backward's items are cut from
        out["visible"] = int((out["radii"] > 0).sum())
    grads = None
    if dL_dpix is not None:
        g = N.rasterize_gaussians_backward(
            args[0], args[1], radii, args[2], args[4], args[5], scene.scale_modifier, args[7], args[8], args[9],
            scene.tanfovx, scene.tanfovy, t(dL_dpix), args[14], scene.sh_degree, args[16], geom, R, binning, img, debug)
        names = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot")
        grads = {n: x.cpu().numpy() for n, x in zip(names, g)}
    return out, grads


def ulp_diff(a, b):
    """Units-in-the-last-place distance between two float32 arrays (same shape)."""
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, np.int64(-2**31) - a, a)
    b = np.where(b < 0, np.int64(-2**31) - b, b)
    return np.abs(a - b)


# ------------------------------------------------------------------------------------------ gradient bars
GRAD_NAMES = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot")
GRAD_ABS = 1e-4     # of max|g| of the tensor
GRAD_REL = 1e-3     # of the element's own reference value
ROW_REL = 2e-3      # per-Gaussian row norm, relative
ROW_ABS = 1e-4      # ... plus this fraction of the largest row norm of the tensor


def grad_violations(a, b):
    """Per-element and per-Gaussian-row comparison of one gradient tensor (a = library, b = reference), in float64:
         element:  |a - b| <= GRAD_ABS * max|b| + GRAD_REL * |b|
         row:      ||a_i - b_i|| <= ROW_REL * ||b_i|| + ROW_ABS * max_j ||b_j||        (row = one Gaussian)
    Returns (number of violating elements, number of violating rows, worst element excess ratio)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64).reshape(a.shape)
    if a.size == 0:
        return 0, 0, 0.0
    scale = np.abs(b).max()
    bound = GRAD_ABS * scale + GRAD_REL * np.abs(b) + 1e-30
    d = np.abs(a - b)
    bad_el = int((d > bound).sum())
    a2, b2 = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    rn = np.linalg.norm(b2, axis=1)
    rd = np.linalg.norm(a2 - b2, axis=1)
    bad_row = int((rd > ROW_REL * rn + ROW_ABS * rn.max() + 1e-30).sum())
    return bad_el, bad_row, float((d / bound).max())


WHOLE_TENSOR = 2e-4  # max|a - b| <= this x max|b| over the whole tensor (round 1's bar, kept next to the finer ones)


def check_grads(gp, go, tag, names=GRAD_NAMES, whole_tensor=WHOLE_TENSOR):
    """Assert every gradient tensor of the library (gp) against a reference (go): finite, same shape, the per-element and
    per-row bars of grad_violations, and the whole-tensor bar max|a - b| <= whole_tensor * max|b| (None: skip it)."""
    for k in names:
        a, b = gp[k], go[k]
        if a.size == 0 and np.asarray(b).size == 0:
            continue
        assert a.size == np.asarray(b).size, (tag, k, a.shape, np.asarray(b).shape)
        assert np.isfinite(a).all(), "%s %s: non-finite gradient" % (tag, k)
        bad_el, bad_row, worst = grad_violations(a, b)
        assert bad_el == 0 and bad_row == 0, "%s %s: %d elements / %d rows outside the bar (worst element at %.2fx its bound)" % (
            tag, k, bad_el, bad_row, worst)
        if whole_tensor is not None:
            b64 = np.asarray(b, np.float64).reshape(a.shape)
            dmax, scale = np.abs(np.asarray(a, np.float64) - b64).max(), np.abs(b64).max()
            assert dmax <= whole_tensor * scale + 1e-30, "%s %s: max|a - b| = %.3g > %.1e * max|b| = %.3g" % (tag, k, dmax, whole_tensor, whole_tensor * scale)


# ---------------------------------------------------------------------------------------------- footprint clipping
def _tile_lists(p):
    T = p["ranges"].shape[0]
    return [p["vals"][p["ranges"][t, 0]:p["ranges"][t, 1]] for t in range(T)]


def check_clipped_equivalent(a, b, tag, check_dead=True):
    """a: a run with the reference's full lists (reference_lists=True), b: the same scene with footprint clipping (the product's
    default).  Everything the API returns must be identical bit for bit; b's lists must be a's with entries REMOVED (same
    order), every removed (tile, Gaussian) pair must be one the reference skips at every pixel of the tile -- replayed here in
    float32 with the reference's expression order (CR/forward.cu:328-347) from the bit-exact per-Gaussian values -- and
    n_contrib must point at the same Gaussian through the shorter lists.  Returns (pairs kept, pairs of the reference)."""
    assert b["R"] == a["R"], tag + ": num_rendered must stay the reference's count"
    np.testing.assert_array_equal(a["radii"], b["radii"], err_msg=tag + " radii")
    assert a["out_color"].tobytes() == b["out_color"].tobytes(), tag + ": out_color differs between full and clipped lists"
    if a["P"] == 0:
        return 0, 0
    assert a["L"] == a["R"] and b["L"] <= a["R"], tag
    if "vals" not in a or "vals" not in b:
        return b["L"], a["R"]
    np.testing.assert_array_equal(a["tiles_touched"], b["tiles_touched"], err_msg=tag + " tiles_touched")
    assert a["final_T"].tobytes() == b["final_T"].tobytes(), tag + ": final_T"
    for k in ("means2D", "depths", "conic_opacity", "rgb"):
        assert a[k].tobytes() == b[k].tobytes(), tag + ": " + k
    W, H = a["W"], a["H"]
    gx = (W + 15) // 16
    la, lb = _tile_lists(a), _tile_lists(b)
    m2, co = a["means2D"].astype(np.float32), a["conic_opacity"].astype(np.float32)
    half, cut = np.float32(-0.5), np.float32(1.0) / np.float32(255.0)
    nca = a["n_contrib"].astype(np.int64)
    ncb = b["n_contrib"].astype(np.int64)
    for t in range(len(la)):
        fa, fb = la[t], lb[t]
        if fa.size == fb.size:
            assert np.array_equal(fa, fb), "%s: tile %d lists differ" % (tag, t)
            pos = np.arange(1, fa.size + 1)
        else:
            # fb must be a subsequence of fa: greedy matching (ids can repeat in neither list: one pair per (tile, Gaussian))
            assert fb.size < fa.size, "%s: tile %d: clipped list longer than the reference's" % (tag, t)
            keep = np.isin(fa, fb)
            assert np.array_equal(fa[keep], fb), "%s: tile %d: clipped list is not the reference's list with entries removed" % (tag, t)
            pos = np.nonzero(keep)[0] + 1              # 1-based reference position of every kept entry
            if check_dead:
                ids = fa[~keep]
                ty, tx = divmod(t, gx)
                xs = np.arange(tx * 16, min(tx * 16 + 16, W), dtype=np.float32)[None, None, :]
                ys = np.arange(ty * 16, min(ty * 16 + 16, H), dtype=np.float32)[None, :, None]
                A, B, C, o = (co[ids, k][:, None, None] for k in range(4))
                dx, dy = m2[ids, 0][:, None, None] - xs, m2[ids, 1][:, None, None] - ys
                power = half * (A * dx * dx + C * dy * dy) - B * dx * dy          # float32 throughout, one rounding per operation
                with np.errstate(over="ignore", under="ignore", invalid="ignore"):
                    alpha = o * np.exp(power)
                live = (~(power > 0)) & ~(alpha < cut * np.float32(1.0 - 1e-5))   # (1e-5: glibc vs ocml expf, last bit)
                assert not live.any(), "%s: tile %d: a removed pair reaches alpha >= 1/255 (Gaussian %d)" % (
                    tag, t, int(ids[np.nonzero(live.any((1, 2)))[0][0]]))
        # n_contrib: b's list position -> the reference's
        ty, tx = divmod(t, gx)
        ya, yb, xa, xb = ty * 16, min(ty * 16 + 16, H), tx * 16, min(tx * 16 + 16, W)
        nb = ncb[ya:yb, xa:xb]
        mapped = np.where(nb > 0, np.concatenate([[0], pos])[np.minimum(nb, pos.size)], 0)
        assert np.array_equal(mapped, nca[ya:yb, xa:xb]), "%s: tile %d: n_contrib does not map back to the reference's positions" % (tag, t)
    return b["L"], a["R"]
