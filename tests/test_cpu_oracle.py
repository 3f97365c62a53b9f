"""CPU tests (no GPU): pin the plain-C oracle.

  * against golden vectors produced by the REFERENCE BUILD (the reference's own kernels compiled for gfx950 and
    run on the MI355X box, tests/golden/ref_*.npz, generator tests/golden/make_golden_ref_gpu.py);
  * against the reference's Python SH evaluator and camera/settings code (tests/golden/py_*.npz, generator
    tests/golden/make_golden_py.py) and the reference's shipped camera fixture;
  * known-answer and property tests of the integer stages (msb, scan, stable sort, ranges).
"""
import hashlib
import os

import numpy as np
import pytest

import util
from util import build_scene, seeded_dL

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_SCENES = sorted(f[4:-4] for f in os.listdir(GOLD) if f.startswith("ref_") and f.endswith(".npz"))


def _digest(s):
    h = hashlib.sha1()
    for f in s.FIELDS:
        a = getattr(s, f)
        h.update(b"-" if a is None else np.ascontiguousarray(a).tobytes())
    h.update(np.array([s.W, s.H, s.sh_degree], np.int64).tobytes())
    h.update(np.array([s.tanfovx, s.tanfovy, s.scale_modifier], np.float64).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("name", REF_SCENES)
def test_oracle_matches_reference_build_golden(name, oracle):
    g = np.load(os.path.join(GOLD, "ref_%s.npz" % name))
    s = build_scene(name)
    assert _digest(s) == str(g["digest"]), "scene generator drifted from the one the golden file was made with"
    dL = seeded_dL(s)
    o, go = oracle.forward_backward(s, dL)
    # integer / index outputs: exact
    assert o["R"] == int(g["R"])
    np.testing.assert_array_equal(o["radii"], g["radii"])
    if s.P == 0 or "keys" not in g:
        np.testing.assert_array_equal(o["out_color"], g["out_color"])
        return
    np.testing.assert_array_equal(o["tiles_touched"], g["tiles_touched"])
    np.testing.assert_array_equal(o["keys"], g["keys"])
    np.testing.assert_array_equal(o["vals"], g["vals"])          # includes the order of tied keys
    np.testing.assert_array_equal(o["ranges"], g["ranges"])
    # per-Gaussian floats: only IEEE +,-,*,/,sqrt -> bit-exact between gcc (no FMA) and hipcc -ffp-contract=off
    vis = o["radii"] > 0
    for k in ("means2D", "depths", "conic_opacity") + (("rgb",) if s.shs is not None else ()):
        assert o[k][vis].tobytes() == g[k].tobytes(), k
    if s.shs is not None:
        np.testing.assert_array_equal(np.packbits(o["clamped"][vis].astype(bool)), g["clamped"])
    # image: glibc expf vs the GPU's expf may differ in the last bit -> tolerance, threshold flips counted
    nc = g["n_contrib"].astype(np.uint32)
    err = np.abs(o["out_color"] - g["out_color"]).max(axis=0)
    flips = (o["n_contrib"] != nc) | (err > 1e-4)
    assert flips.mean() <= 2e-3, "%d flip pixels" % flips.sum()
    assert err[~flips].max(initial=0.0) <= 1e-4          # north_star tolerance, fp32 max abs
    # gradients (reference = float atomics in arbitrary order, oracle = double accumulation)
    if not flips.any():
        for k in ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot"):
            a, b = go[k].astype(np.float64).ravel(), g[k].astype(np.float64).ravel()
            if b.size:
                assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max() + 1e-30, k


def test_sh_colour_matches_python_reference(oracle):
    """computeColorFromSH (CR/forward.cu:20-71) vs models/sh_utils.py::eval_sh of the reference."""
    g = np.load(os.path.join(GOLD, "py_sh_eval.npz"))
    for deg in range(4):
        sh, dirs, want = g["deg%d_sh" % deg], g["deg%d_dirs" % deg], g["deg%d_result" % deg]
        K = (deg + 1) ** 2
        for i in range(sh.shape[0]):
            coeffs = np.ascontiguousarray(sh[i].T)               # [K,3]: coefficient-major rows like the rasterizer
            rgb, clamped = oracle.sh_to_rgb(deg, dirs[i] * 2.5, np.zeros(3, np.float32), coeffs)
            ref = want[i] + 0.5
            np.testing.assert_allclose(rgb, np.maximum(ref, 0.0), atol=2e-6, rtol=1e-5)
            np.testing.assert_array_equal(clamped, ref < 0) if np.abs(ref).min() > 1e-5 else None
        assert K == coeffs.shape[0]
    from pcrender import synth
    np.testing.assert_allclose((g["rgb"] - 0.5) / synth.SH_C0, g["rgb2sh"], rtol=1e-6)


def test_camera_and_settings_match_python_reference_and_fixture():
    from pcrender import camera
    g = np.load(os.path.join(GOLD, "py_camera.npz"))
    Hs = camera.circle_path(12, 0, 3, [90, 0]).numpy()
    assert np.array_equal(Hs, g["fixture_H_c2w"][0])            # the reference's validate/temp_state_dict.pt
    f = 0.5 * 512 / np.tan(0.5 * 45.0 / 180.0 * np.pi)          # plib/render.py:463
    np.testing.assert_allclose(g["fixture_intrinsic"][0, 0], [[f, 0, 256], [0, f, 256], [0, 0, 1]], rtol=1e-6)
    for tag in ("native", "hd", "fov60"):
        w, h, fov, ss = g[tag + "_args"]
        views = camera.circle_views(12, fov_deg=float(fov), width_px=int(w), height_px=int(h), super_sample_rate=int(ss))
        for i, v in enumerate(views):
            assert np.array_equal(v["viewmatrix"].numpy(), g[tag + "_viewmatrix"][i])
            assert np.array_equal(v["projmatrix"].numpy(), g[tag + "_projmatrix"][i])
            assert np.array_equal(v["campos"].numpy(), g[tag + "_campos"][i])
            assert v["tanfovx"] == g[tag + "_tanfov"][i, 0] and v["tanfovy"] == g[tag + "_tanfov"][i, 1]
            assert [v["image_height"], v["image_width"]] == list(g[tag + "_size"][i])
    # quirk Q1: tan of the FULL angle goes to the rasterizer, the projection uses the half angle
    v = camera.circle_views(12, fov_deg=45.0, width_px=512, height_px=512, super_sample_rate=2)[1]
    assert abs(v["tanfovx"] - 1.0) < 1e-12 and abs(float(v["projmatrix"][1, 1]) + 2.4142137) < 1e-6


def test_get_higher_msb_known_answers(oracle):
    # CR/rasterizer_impl.cu:35-50: floor(log2 n) + 1; SURVEY App. A.5: 4096 -> 13, 8160 -> 13, 32400 -> 15
    for n, want in [(1, 1), (2, 2), (3, 2), (4, 3), (255, 8), (256, 9), (4096, 13), (8160, 13), (32400, 15), (65535, 16)]:
        assert oracle.get_higher_msb(n) == want


def test_sort_pairs_is_stable_and_masks_high_bits(oracle):
    rng = np.random.default_rng(5)
    n = 20000
    keys = (rng.integers(0, 37, n).astype(np.uint64) << np.uint64(32)) | rng.integers(0, 50, n).astype(np.uint64)
    vals = np.arange(n, dtype=np.uint32)
    ko, vo = oracle.sort_pairs(keys, vals, 32 + 6)
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(ko, keys[order])
    np.testing.assert_array_equal(vo, vals[order])            # ties keep emission order
    # bits at or above end_bit are ignored by the comparison (cub SortPairs(begin_bit, end_bit) semantics)
    k2 = keys | (rng.integers(0, 2, n).astype(np.uint64) << np.uint64(60))
    ko2, vo2 = oracle.sort_pairs(k2, vals, 32 + 6)
    np.testing.assert_array_equal(vo2, vo)


def test_forward_properties(oracle):
    s = build_scene("big_splats")
    o = oracle.forward(s)
    # ranges partition [0,R) in tile order; every list is sorted by (depth bits, id)
    r = o["ranges"]
    nz = r[r[:, 1] > r[:, 0]]
    assert nz[0, 0] == 0 and nz[-1, 1] == o["R"] and (nz[1:, 0] == nz[:-1, 1]).all()
    assert (np.diff(o["keys"].astype(np.uint64)) >= 0).all() if o["R"] > 1 else True
    assert o["point_offsets"][-1] == o["R"] == o["tiles_touched"].sum()
    # compositing is affine in the background: out(bg) = out(0) + final_T * bg
    s0 = build_scene("big_splats"); s0.bg[:] = 0
    o0 = oracle.forward(s0)
    np.testing.assert_allclose(o["out_color"], o0["out_color"] + o["final_T"][None] * s.bg[:, None, None], atol=1e-6)
    # transmittance in [1e-4*(1-0.99), 1]; n_contrib never exceeds the tile's list length
    assert o["final_T"].min() > 0 and o["final_T"].max() <= 1.0
    assert o["consumed_bwd"] <= o["consumed_fwd"] <= o["R"]
    # threads: the OpenMP build gives the same forward whatever the thread count
    o4 = oracle.forward(s, nthreads=4)
    assert o4["out_color"].tobytes() == o["out_color"].tobytes() and np.array_equal(o4["vals"], o["vals"])


def test_threaded_scan_and_sort_survive_a_smaller_team():
    """the oracle's parallel scan / radix sort cut their input into `nthreads` chunks; when the OpenMP runtime
    grants a smaller team (OMP_THREAD_LIMIT, dynamic teams) every chunk must still be scanned and scattered.  Runs in a child
    process (the limit has to be in the environment before libgomp starts): 70 000 Gaussians (>= 65 536: the threaded scan)
    and > 65 536 pairs (the threaded sort), 8 chunks on a team of at most 3, against the single-thread result."""
    import subprocess, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        "sys.path[:0] = [%r, %r, %r]\n"
        "import util\n"
        "from oracle.oracle import Oracle\n"
        "from pcrender import synth\n"
        "g = synth.random_scene(70000, 96, 64, seed=5, sh_degree=0, spread=1.0, scale=0.01)\n"
        "s = util.scene_from(g, util.identity_camera(96, 64), 96, 64, bg=(0, 0, 0))\n"
        "o = Oracle()\n"
        "a, b = o.forward(s, nthreads=1), o.forward(s, nthreads=8)\n"
        "assert a['R'] == b['R'] >= 65536, a['R']\n"
        "for k in ('point_offsets', 'keys', 'vals', 'ranges', 'n_contrib'):\n"
        "    assert np.array_equal(a[k], b[k]), k\n"
        "assert a['out_color'].tobytes() == b['out_color'].tobytes()\n"
        "print('ok', a['R'])\n") % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd"))
    env = dict(os.environ, OMP_THREAD_LIMIT="3", OMP_DYNAMIC="true")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr


def test_empty_and_degenerate_inputs(oracle):
    o = oracle.forward(build_scene("all_culled"))
    s = build_scene("all_culled")
    assert o["R"] == 0 and not o["radii"].any()
    np.testing.assert_array_equal(o["out_color"], np.broadcast_to(s.bg[:, None, None], o["out_color"].shape))
    # P == 0: zero image, NOT the background (rasterize_points.cu:68,81)
    from oracle.oracle import Scene
    e = Scene(W=20, H=10, tanfovx=1, tanfovy=1, bg=[1, 1, 1], means3D=np.zeros((0, 3)), opacities=np.zeros((0,)),
              viewmatrix=np.eye(4), projmatrix=np.eye(4), campos=np.zeros(3), colors_precomp=np.zeros((0, 3)),
              scales=np.zeros((0, 3)), rotations=np.zeros((0, 4)))
    o = oracle.forward(e)
    assert o["R"] == 0 and not o["out_color"].any()


def test_gradients_against_finite_differences(oracle):
    """Central differences of loss = sum(out * G) in float64-accumulated fp32 renders on a tiny, well-conditioned
    scene (coarse check of the analytic backward's signs and scales, not of its last digits)."""
    from oracle.oracle import Scene
    rng = np.random.default_rng(3)
    P, W, H = 6, 32, 24
    view = util.identity_camera(W, H)
    base = dict(W=W, H=H, tanfovx=view["tanfovx"], tanfovy=view["tanfovy"], bg=[0.1, 0.2, 0.3],
                viewmatrix=np.asarray(view["viewmatrix"]), projmatrix=np.asarray(view["projmatrix"]),
                campos=np.asarray(view["campos"]), sh_degree=1)
    par = dict(means3D=np.stack([rng.uniform(-.3, .3, P), rng.uniform(-.2, .2, P), rng.uniform(1.5, 2.5, P)], 1),
               scales=rng.uniform(0.15, 0.3, (P, 3)), rotations=rng.standard_normal((P, 4)) * 0.3 + [1, 0, 0, 0],
               opacities=rng.uniform(0.3, 0.6, (P, 1)), shs=rng.standard_normal((P, 4, 3)) * 0.3)
    G = rng.uniform(-1, 1, (3, H, W)).astype(np.float32)

    def loss(p):
        o = oracle.forward(Scene(**base, **{k: v.astype(np.float32) for k, v in p.items()}))
        return float((o["out_color"].astype(np.float64) * G).sum())

    s = Scene(**base, **{k: v.astype(np.float32) for k, v in par.items()})
    _, g = oracle.forward_backward(s, G)
    names = dict(means3D="dL_dmean3D", scales="dL_dscale", rotations="dL_drot", opacities="dL_dopacity", shs="dL_dsh")
    for key, gname in names.items():
        an = g[gname].reshape(par[key].shape)
        for idx in [(0, 0), (2, 1), (5, 2)] if par[key].ndim == 2 and par[key].shape[1] > 2 else [(0, 0), (3, 0)]:
            idx = idx + (0,) * (par[key].ndim - 2)
            eps = 2e-3
            pp = {k: v.copy() for k, v in par.items()}
            pm = {k: v.copy() for k, v in par.items()}
            pp[key][idx] += eps
            pm[key][idx] -= eps
            fd = (loss(pp) - loss(pm)) / (2 * eps)
            assert abs(fd - an[idx]) <= 0.05 * max(abs(fd), abs(an[idx])) + 0.02, (key, idx, fd, an[idx])


@pytest.mark.parametrize("name", ["random_aniso", "sh_deg2", "sh_deg3", "cov3d_precomp", "scale_modifier", "capsule_circle"])
def test_fp64_gaussian_backward_agrees_with_the_oracle(name, oracle):
    """tests/fp64_backward.py (the float64 arbiter of the randomised sweep, written from the formulas, not from the
    reference's expression list) against the float32 oracle on well-conditioned scenes."""
    import util
    from fp64_backward import gaussian_backward_fp64
    s = util.build_scene(name)
    f, g = oracle.forward_backward(s, util.seeded_dL(s))
    t = gaussian_backward_fp64(s, f["radii"], f["clamped"], g["dL_dmean2D"], g["dL_dconic"], g["dL_dcolor"])
    assert set(t) >= {"dL_dmean3D", "dL_dcov3D"}
    for k, v in t.items():
        a = g[k].astype(np.float64).reshape(v.shape)
        assert np.abs(a - v).max() <= 5e-6 * np.abs(v).max() + 1e-12, k
