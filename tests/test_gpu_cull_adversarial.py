"""Adversarial test of the exact-safe footprint cull (csrc/tile_cull.hpp): splats placed so that the LARGEST power over a
quadrant's 64 pixel centres sits within a hair of log(1 / (255 * opacity)) -- the value at which the reference starts to
skip the entry for a pixel (alpha < 1/255, CR/forward.cu:336-347) -- approached over every edge and corner of each quadrant
of a tile, with axis-aligned, anisotropic and rotated footprints.  If the cull's rounding allowance E
(1e-5 * terms + 1e-4 + 1e-5 * |thr|) were too small somewhere, a splat whose alpha just reaches 1/255 at one pixel would be
dropped for the quadrant and the image / n_contrib would differ from the reference build's, which evaluates every entry at
every pixel.  The test first proves that the sweep really brackets the threshold on both sides, closely."""
import numpy as np
import pytest

import util
from util import run_product

pytestmark = pytest.mark.gpu

TILES_X, TILES_Y = 42, 38          # one splat per interior tile
W, H = TILES_X * 16, TILES_Y * 16
Z0 = 2.0


def _ref():
    return util.reference_build("strict")


def _targets():
    """(tile, quadrant, approach) for every tile: approach = unit direction from the quadrant towards the splat's mean and the
    quadrant pixel it is nearest to (edge midpoints, slightly off-centre edge points, corners)."""
    out = []
    appr = []
    for qx in (0, 1):
        for qy in (0, 1):
            x0, y0 = 8 * qx, 8 * qy
            for (ax, ay, dx, dy) in (
                    (x0, y0 + 3, -1, 0), (x0 + 7, y0 + 4, 1, 0), (x0 + 3, y0, 0, -1), (x0 + 4, y0 + 7, 0, 1),       # edges
                    (x0, y0, -1, -1), (x0 + 7, y0, 1, -1), (x0, y0 + 7, -1, 1), (x0 + 7, y0 + 7, 1, 1),             # corners
                    (x0, y0 + 6, -1, 0.15), (x0 + 5, y0 + 7, 0.2, 1)):                                               # oblique
                n = float(np.hypot(dx, dy))
                appr.append((qx, qy, ax, ay, dx / n, dy / n))
    k = 0
    for ty in range(1, TILES_Y - 1):          # (the border tiles stay empty: a mean pushed off the image would be another test)
        for tx in range(1, TILES_X - 1):
            out.append((tx, ty) + appr[k % len(appr)] + (k // len(appr),))
            k += 1
    return out, len(appr)


def _scene(variant):
    """variant 0: isotropic, 1: anisotropic axis-aligned, 2: anisotropic rotated about the view axis"""
    view = util.identity_camera(W, H, 60.0)
    fx = W / (2.0 * view["tanfovx"])
    fy = H / (2.0 * view["tanfovy"])
    # ndc2Pix: pix = ((x_ndc + 1) W - 1) / 2 with x_ndc = proj . p / w; at the identity camera this is fx' x / z + (W - 1) / 2 with
    # fx' = W / 2 * proj[0]; measured below from the oracle rather than derived (the reference's tan / fov quirk, Q1)
    targets, n_appr = _targets()
    P = len(targets)
    rng = np.random.default_rng(5 + variant)
    opac = np.full((P, 1), 0.30, np.float32)
    thr = -np.log(255.0 * 0.30)
    sig_px = 3.0
    s_world = sig_px * Z0 / fx
    scales = np.full((P, 3), s_world, np.float32)
    rots = np.tile(np.array([1, 0, 0, 0], np.float32), (P, 1))
    if variant >= 1:
        scales[:, 0] *= 1.35
        scales[:, 1] *= 0.8
    if variant == 2:
        ang = rng.uniform(0, np.pi, P)
        rots = np.stack([np.cos(ang / 2), np.zeros(P), np.zeros(P), np.sin(ang / 2)], 1).astype(np.float32)
    g = dict(means3D=np.zeros((P, 3), np.float32), scales=scales, rotations=rots, opacities=opac,
             shs=(0.4 * rng.standard_normal((P, 4, 3))).astype(np.float32), sh_degree=1)
    return g, view, targets, n_appr, thr, fx, fy


def _quadrant_max_power(m2, co, targets):
    """largest power over the 64 pixel centres of each splat's targeted quadrant, float64"""
    out = np.empty(len(targets))
    for i, t in enumerate(targets):
        x0, y0 = 16 * t[0] + 8 * t[2], 16 * t[1] + 8 * t[3]
        xs, ys = np.meshgrid(np.arange(x0, x0 + 8), np.arange(y0, y0 + 8))
        dx, dy = m2[i, 0] - xs, m2[i, 1] - ys
        out[i] = (-0.5 * (co[i, 0] * dx * dx + co[i, 2] * dy * dy) - co[i, 1] * dx * dy).max()
    return out


def _place(g, view, targets, n_appr, thr, fx, fy, oracle):
    """Put every mean on the ray from its anchor pixel along the approach direction, at the distance where the LARGEST power over
    the targeted quadrant equals thr * (1 + delta).  The 2D conic depends (slightly) on where the splat sits, so the distance is
    refined a few times against what the pipeline really produces (the oracle's means2D / conic)."""
    P = len(targets)
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    anchor = np.array([[16 * t[0] + t[4], 16 * t[1] + t[5]] for t in targets], np.float64)
    dirs = np.array([[t[6], t[7]] for t in targets], np.float64)
    # relative offsets of the target power around thr: dense within +-3e-4 (the cull's allowance E is ~1.5e-4 here), a few further out
    n_delta = (P + n_appr - 1) // n_appr
    deltas = np.concatenate([np.linspace(-3e-4, 3e-4, n_delta - 8), [-3e-3, -1e-3, 1e-3, 3e-3, -2e-2, 2e-2, -1e-6, 1e-6]])
    delta = np.array([deltas[t[8] % len(deltas)] for t in targets])
    target = thr * (1.0 + delta)                  # negative
    # first guess of the pixel <-> world map: the projection matrix maps with the HALF-angle tangent while the covariance uses
    # focal = W / (2 tan(full angle)) (the reference's quirk Q1), so the two focal lengths differ; refined from the pipeline's
    # own means2D after the first pass
    pm = np.asarray(view["projmatrix"], np.float64).reshape(4, 4)
    kx, ky = np.array([0.5 * W * pm[0, 0], cx]), np.array([0.5 * H * pm[1, 1], cy])
    tdist = np.full(P, 9.0)
    s = None
    for it in range(6):
        px, py = anchor[:, 0] + tdist * dirs[:, 0], anchor[:, 1] + tdist * dirs[:, 1]
        g["means3D"] = np.stack([((px - kx[1]) / kx[0]) * Z0, ((py - ky[1]) / ky[0]) * Z0, np.full(P, Z0)], 1).astype(np.float32)
        s = util.scene_from(g, view, W, H, bg=(0.1, 0.2, 0.3))
        o = oracle.forward(s)
        m2 = o["means2D"].astype(np.float64)
        if it == 0:   # the pixel <-> world map as the pipeline applies it (the reference's tan / fov quirk, Q1): affine per axis
            kx = np.polyfit(g["means3D"][:, 0] / Z0, m2[:, 0], 1)
            ky = np.polyfit(g["means3D"][:, 1] / Z0, m2[:, 1], 1)
        pmax = _quadrant_max_power(m2, o["conic_opacity"].astype(np.float64), targets)
        # distance actually realised along the ray, and the rescaling that would bring pmax to the target if power ~ -t^2
        t_real = (m2[:, 0] - anchor[:, 0]) * dirs[:, 0] + (m2[:, 1] - anchor[:, 1]) * dirs[:, 1]
        tdist = np.where(pmax < 0, t_real * np.sqrt(target / np.minimum(pmax, -1e-9)), tdist * 1.5) + (tdist - t_real)
    return s, anchor, delta


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_splats_grazing_the_alpha_cut_on_every_quadrant_edge(variant, oracle, gpu_device):
    ref = _ref()
    g, view, targets, n_appr, thr, fx, fy = _scene(variant)
    s, anchor, delta = _place(g, view, targets, n_appr, thr, fx, fy, oracle)
    r = ref.forward(s)
    # --- how close to the cut did the sweep get?  margin = (max power over the targeted quadrant's pixels) - thr, in float64
    # from the reference build's own 2D means / conics
    m2, co = r["means2D"].astype(np.float64), r["conic_opacity"].astype(np.float64)
    margins = _quadrant_max_power(m2, co, targets) - (-np.log(255.0 * co[:, 3]))
    vis = r["radii"] > 0
    assert vis.all()
    just_in = int(((margins >= 0) & (margins < 2e-4)).sum())
    just_out = int(((margins < 0) & (margins > -2e-4)).sum())
    hair = int((np.abs(margins) < 2e-5).sum())
    print("variant %d: %d splats; margins within 2e-4 of the cut: %d inside, %d outside; within 2e-5: %d; closest %.2e"
          % (variant, len(targets), just_in, just_out, hair, np.abs(margins).min()))
    assert just_in >= 100 and just_out >= 100 and hair >= 40, "the sweep does not bracket the threshold closely enough to mean anything"
    # the reference build really does see both sides: some of these splats contribute to their quadrant, some to no pixel of it
    # --- the product against the reference build: bit for bit
    p, _ = run_product(s, gpu_device)
    assert p["R"] == r["R"]
    np.testing.assert_array_equal(p["n_contrib"], r["n_contrib"])
    assert p["final_T"].tobytes() == r["final_T"].tobytes()
    assert p["out_color"].tobytes() == r["out_color"].tobytes()
    # and the backward's cull (it tests against the pixels that consumed that deep) leaves the gradients where they belong
    dL = util.seeded_dL(s)
    _, gr = ref.forward_backward(s, dL)
    _, gp = run_product(s, gpu_device, dL_dpix=dL)
    gr = dict(gr)
    gr["dL_dopacity"] = gr["dL_dopacity"].reshape(gp["dL_dopacity"].shape)
    util.check_grads(gp, gr, "grazing splats, variant %d" % variant)
    # --- the same sweep against footprint clipping (the product's default list construction, csrc/tile_cull.hpp
    # clip_rect_to_footprint): a quarter of the targeted quadrants lie on their tile's left / top edge with the splat's mean in the
    # neighbouring tile, so the clipped rectangle's edge is decided by the very pixel column / row the sweep grazes
    c, gc = run_product(s, gpu_device, dL_dpix=dL, reference_lists=False)
    kept, total = util.check_clipped_equivalent(p, c, "grazing splats, variant %d, clipped lists" % variant)
    assert c["final_T"].tobytes() == r["final_T"].tobytes() and c["out_color"].tobytes() == r["out_color"].tobytes()
    assert kept < total, "no pair was clipped: the sweep does not exercise the clipping"
    util.check_grads(gc, gr, "grazing splats, variant %d, clipped lists" % variant)
    print("variant %d: clipped lists keep %d of %d pairs" % (variant, kept, total))
