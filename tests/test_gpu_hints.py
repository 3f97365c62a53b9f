"""Depth hints (C ABI gsr_forward_batch_hinted, include/gsr.h): the lists of a tile are cut where the previous call with the same
hint buffer stopped consuming them.  A hint is never trusted -- a cut list that is outrun makes the call repeat its binning half
without the filter -- so every test here demands BIT-IDENTICAL API outputs (image, radii, num_rendered), identical final_T /
n_contrib, and the backward's gradients, against the unhinted call and the reference build, whatever happens to the geometry between
calls."""
import numpy as np
import pytest
import torch

import util
from util import build_scene, seeded_dL

pytestmark = pytest.mark.gpu


def _ref():
    from oracle.oracle import Reference
    if not Reference.available("strict"):
        pytest.skip("oracle/_ref not built")
    return Reference("strict")


def _wall(opacity=1.0, seed=5):
    """4 000 large splats of the given opacity in front of a 96 x 96 camera: with opacity 1 every pixel stops after two or three list
    entries while the tiles' lists hold ~1 000: the scene depth hints are for"""
    from pcrender import synth
    W = H = 96
    g = synth.random_scene(4000, W, H, seed=seed, sh_degree=1, spread=1.0, scale=0.4)
    g["opacities"][:] = opacity
    return util.scene_from(g, util.identity_camera(W, H, 60.0), W, H, bg=(0.2, 0.4, 0.6))


def _scene(name):
    return _wall() if name == "wall" else build_scene(name)


def _args(s, dev):
    def t(a):
        return torch.empty(0) if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return (t(s.bg), t(s.means3D), t(s.colors_precomp), t(s.opacities), t(s.scales), t(s.rotations), s.scale_modifier, t(s.cov3D_precomp),
            t(s.viewmatrix.reshape(1, 4, 4)), t(s.projmatrix.reshape(1, 4, 4)), s.tanfovx, s.tanfovy, s.H, s.W, t(s.shs), s.sh_degree,
            t(s.campos.reshape(1, 3)), s.prefiltered, False)


def _call(N, s, dev, dL=None, hint=None, slack=0.02):
    a = _args(s, dev)
    counts, color, radii, geom, binning, img = N.rasterize_gaussians_batch(*a, need_backward=dL is not None, depth_hint=hint,
                                                                         hint_slack=slack)
    R = counts[0]
    q = lambda name: N.query(name, s.P, s.W, s.H, R, geom, binning, img).cpu().numpy()   # noqa: E731
    out = dict(R=R, color=color[0].cpu().numpy(), radii=radii[0].cpu().numpy(), final_T=q("FINAL_T"), n_contrib=q("N_CONTRIB"),
               ranges=q("RANGES").view(np.uint32), need=q("TILE_NEED"))
    grads = None
    if dL is not None:
        g = N.rasterize_gaussians_backward_batch(a[0], a[1], radii, a[2], a[4], a[5], s.scale_modifier, a[7], a[8], a[9], s.tanfovx,
                                                 s.tanfovy, torch.from_numpy(dL).to(dev)[None], a[14], s.sh_degree, a[16], geom, binning,
                                                 img, False)
        grads = [x.cpu().numpy() for x in g]
    return out, grads


def _same(a, b, what):
    for k in ("R", "color", "radii", "final_T", "n_contrib"):
        x, y = a[k], b[k]
        assert (x == y) if np.isscalar(x) else (x.tobytes() == y.tobytes()), (what, k)


def _grads_close(ga, gb, what):
    for i, (x, y) in enumerate(zip(ga, gb)):
        if x.size:
            assert np.abs(x.astype(np.float64) - y).max() <= 2e-4 * (np.abs(y).max() + 1e-30), (what, i)   # (two runs differ by the order their float atomics commit in)


@pytest.mark.parametrize("name", ["wall", "opaque_early_stop", "deep_stack", "capsule_circle", "voxel_ties", "culled_mix"])
def test_repeated_view_renders_identically_with_shorter_lists(name, gpu_device):
    from diff_gaussian_rasterization import _native as N
    s = _scene(name)
    dL = seeded_dL(s)
    plain, gplain = _call(N, s, gpu_device, dL)
    ref = _ref().forward(s)
    assert plain["color"].tobytes() == ref["out_color"].tobytes()
    hint = N.new_depth_hints(1, s.W, s.H, gpu_device)
    st0 = N.hint_stats()
    first, g1 = _call(N, s, gpu_device, dL, hint=hint)          # nothing to cut yet; leaves hints behind
    _same(first, plain, name + " first hinted call")
    assert (first["ranges"] == plain["ranges"]).all()
    h = hint.cpu().numpy().view(np.uint32)[0]
    sat = h != 0xFFFFFFFF
    for rep in range(2):                                        # the hints now cut the saturated tiles' lists
        again, g2 = _call(N, s, gpu_device, dL, hint=hint)
        _same(again, plain, name + " hinted call %d" % rep)
        _grads_close(g2, gplain, name)
        lens_plain = (plain["ranges"][:, 1] - plain["ranges"][:, 0]).astype(np.int64)
        lens_hint = (again["ranges"][:, 1] - again["ranges"][:, 0]).astype(np.int64)
        assert (lens_hint <= lens_plain).all() and (lens_hint >= plain["need"]).all()   # a prefix that still holds what was consumed
        assert (lens_hint[~sat] == lens_plain[~sat]).all()      # tiles that did not saturate keep everything
        assert (hint.cpu().numpy().view(np.uint32)[0] == h).all()      # same scene: the hints reproduce themselves
    st1 = N.hint_stats()
    assert st1["hinted_forwards"] - st0["hinted_forwards"] == 3 and st1["repeated"] == st0["repeated"]
    if name == "wall":
        assert sat.sum() >= 0.5 * sat.size and lens_hint.sum() < 0.2 * lens_plain.sum(), (int(sat.sum()), int(lens_hint.sum()), int(lens_plain.sum()))
    if sat.sum():
        print("%s: %d of %d non-empty tiles saturated; list pairs %d -> %d" % (name, int(sat.sum()), int((lens_plain > 0).sum()),
                                                                             int(lens_plain.sum()), int(lens_hint.sum())))


def test_outrun_hints_repeat_the_frame_and_match_the_reference_build(gpu_device):
    """Hints taken on an opaque cloud (every covered tile saturates after a few entries), then the cloud turns nearly transparent
    (opacity x 0.04: nothing saturates any more, every cut list is outrun): the call repeats its binning half without the filter
    and returns exactly what the reference build renders; the next call on the new cloud has hints that fit it."""
    from diff_gaussian_rasterization import _native as N
    import copy
    s = _wall()
    hint = N.new_depth_hints(1, s.W, s.H, gpu_device)
    _call(N, s, gpu_device, hint=hint)
    h = hint.cpu().numpy().view(np.uint32)[0]
    covered = int((h != 0xFFFFFFFF).sum())
    T = h.size
    assert covered >= 0.1 * T, (covered, T)                     # the fallback below fires on at least a tenth of all tiles
    s2 = copy.copy(s)
    s2.opacities = (s.opacities * 0.04).astype(np.float32)
    dL = seeded_dL(s2)
    plain, gplain = _call(N, s2, gpu_device, dL)
    r, gr = _ref().forward_backward(s2, dL)
    assert plain["color"].tobytes() == r["out_color"].tobytes()
    st0 = N.hint_stats()
    got, g = _call(N, s2, gpu_device, dL, hint=hint)
    st1 = N.hint_stats()
    assert st1["repeated"] == st0["repeated"] + 1               # the cut lists were outrun
    _same(got, plain, "outrun hints")
    assert got["color"].tobytes() == r["out_color"].tobytes() and got["final_T"].tobytes() == r["final_T"].reshape(-1).tobytes()
    _grads_close(g, gplain, "outrun hints")
    assert (got["ranges"] == plain["ranges"]).all()             # the repeat built the full lists
    # and the hints it left fit the transparent cloud: no repeat on the next call, same image
    got2, _ = _call(N, s2, gpu_device, dL, hint=hint)
    assert N.hint_stats()["repeated"] == st1["repeated"]
    _same(got2, plain, "after the repeat")


def test_geometry_drift_between_calls(gpu_device):
    """A training-like sequence: the cloud moves a little between calls (positions, opacities), the hint buffer is carried along;
    every call equals the unhinted render of ITS cloud bit for bit, with or without a repeat."""
    from diff_gaussian_rasterization import _native as N
    import copy
    s = _wall(opacity=0.9)
    rng = np.random.default_rng(3)
    hint = N.new_depth_hints(1, s.W, s.H, gpu_device)
    st0 = N.hint_stats()
    cut_pairs = full_pairs = 0
    for step in range(8):
        s = copy.copy(s)
        s.means3D = (s.means3D + 0.004 * rng.standard_normal(s.means3D.shape)).astype(np.float32)
        s.opacities = np.clip(s.opacities * (1 + 0.05 * rng.standard_normal(s.opacities.shape)), 0.02, 1.0).astype(np.float32)
        plain, _ = _call(N, s, gpu_device)
        got, _ = _call(N, s, gpu_device, hint=hint, slack=0.05)
        _same(got, plain, "drift step %d" % step)
        cut_pairs += int((got["ranges"][:, 1] - got["ranges"][:, 0]).sum())
        full_pairs += int((plain["ranges"][:, 1] - plain["ranges"][:, 0]).sum())
    st1 = N.hint_stats()
    print("drift: %d hinted forwards, %d repeats, list pairs %d of %d" % (st1["hinted_forwards"] - st0["hinted_forwards"],
                                                                          st1["repeated"] - st0["repeated"], cut_pairs, full_pairs))
    assert cut_pairs <= full_pairs


def test_hinted_batch_of_views(gpu_device):
    """V = 3 views in one submission (the 8 x 8 forward kernel; single views run the half-quadrant one), one hint row per view"""
    from diff_gaussian_rasterization import _native as N
    from pcrender import camera, synth
    dev = gpu_device
    g = synth.make_gaussians(synth.make_cloud("synth-THuman-256", seed=0, P=30000), profile="inference", seed=1)
    W, H = 208, 160
    views = camera.circle_views(n_imgs=12, fov_deg=45.0, width_px=W, height_px=H)[:3]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    e = torch.empty(0)
    vm = torch.stack([v["viewmatrix"] for v in views]).to(dev)
    pm = torch.stack([v["projmatrix"] for v in views]).to(dev)
    cp = torch.stack([v["campos"] for v in views]).to(dev)
    args = (torch.ones(3, device=dev), t(g["means3D"]), e, t(g["opacities"]), t(g["scales"]), t(g["rotations"]), 1.0, e, vm, pm,
            views[0]["tanfovx"], views[0]["tanfovy"], H, W, t(g["shs"]), g["sh_degree"], cp, False, False)
    c0, col0, rad0, *_ = N.rasterize_gaussians_batch(*args, need_backward=False)
    hint = N.new_depth_hints(3, W, H, dev)
    st0 = N.hint_stats()
    for k in range(3):
        c, col, rad, geom, binning, img = N.rasterize_gaussians_batch(*args, need_backward=False, depth_hint=hint)
        assert c == c0 and torch.equal(col, col0) and torch.equal(rad, rad0), k
    assert N.hint_stats()["repeated"] == st0["repeated"]
    assert int((hint != -1).sum()) > 0


def test_opt_in_switch_through_the_public_api(gpu_device):
    """GSR_DEPTH_HINT / set_depth_hints(True): rasterize_views on a settings list it has seen before takes hints on its own (keyed by
    the cached view block), the per-view GaussianRasterizer call on prebuilt settings too; images and gradients equal the hints-off
    run; off by default."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, rasterize_views, _native as N
    dev = gpu_device
    s = _wall(opacity=0.95)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    st = GaussianRasterizationSettings(image_height=s.H, image_width=s.W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=t(s.bg), scale_modifier=1.0,
                                       viewmatrix=t(s.viewmatrix.reshape(4, 4)), projmatrix=t(s.projmatrix.reshape(4, 4)), sh_degree=s.sh_degree,
                                       campos=t(s.campos), prefiltered=False, debug=False)
    G = t(seeded_dL(s))

    def leaves():
        L = dict(means3D=t(s.means3D).requires_grad_(True), shs=t(s.shs).requires_grad_(True), opacities=t(s.opacities).requires_grad_(True),
                 scales=t(s.scales).requires_grad_(True), rotations=t(s.rotations).requires_grad_(True))
        L["means2D"] = torch.zeros_like(L["means3D"], requires_grad=True)
        return L

    def run(n):
        L = leaves()
        imgs = []
        for k in range(n):
            if k % 2 == 0:
                img, _ = GaussianRasterizer(st)(**L)
            else:
                img, _ = rasterize_views(L["means3D"], L["means2D"], L["opacities"], [st, st], shs=L["shs"], scales=L["scales"],
                                         rotations=L["rotations"])
                img = img[1]
            (img * G).sum().backward()
            imgs.append(img.detach().clone())
        return imgs, {k: v.grad.clone() for k, v in L.items()}

    assert N._DEPTH_HINTS_ON is False
    want, gw = run(6)
    st0 = N.hint_stats()
    N.set_depth_hints(True)
    try:
        got, gg = run(6)
    finally:
        N.set_depth_hints(False)
    st1 = N.hint_stats()
    assert st1["hinted_forwards"] - st0["hinted_forwards"] == 6 and st1["repeated"] == st0["repeated"]
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    for k in gw:
        assert float((gg[k] - gw[k]).abs().max()) <= 2e-4 * (float(gw[k].abs().max()) + 1e-30), k
