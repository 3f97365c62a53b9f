"""Host-side restatement of the algebra behind the render backward's matrix-core reduction (csrc/render_bwd.hip, "wave
reduction on the matrix cores"): the nine per-entry gradients of reference CR/backward.cu:478-531 are sums over the pixels of a
quadrant of two weights per (pixel, entry), q = G * dL_dalpha and u = alpha * T; the kernel takes the moments of q about the
QUADRANT centre with a constant basis and shifts them to the splat's centre per entry.  This checks, in float64 numpy, that
the shifted moments equal the reference's per-pixel sums, that the LDS row order (2x2 pixel blocks, one block per K step of
v_mfma_f32_16x16x4_f32) is the permutation the basis assumes, and that the hand-over of an entry's nine values to nine lanes
addresses every component exactly once.  (The kernel itself is compared with the oracle and the reference build in the -m gpu
tests; its operand layout by gsr_selftest on the device.)"""
import numpy as np


def mm_pos(lane):
    x, y = lane & 7, lane >> 3
    return 16 * (y >> 1) + 4 * ((x & 1) + 2 * (y & 1)) + (x >> 1)


def test_row_order_puts_one_2x2_block_in_every_k_step():
    pos = np.array([mm_pos(l) for l in range(64)])
    assert sorted(pos.tolist()) == list(range(64))
    for s in range(16):                      # step s = 4 m + r reads row positions 16 m + 4 k + r, k = 0..3
        m, r = s >> 2, s & 3
        lanes = [int(np.nonzero(pos == 16 * m + 4 * k + r)[0][0]) for k in range(4)]
        xs, ys = [l & 7 for l in lanes], [l >> 3 for l in lanes]
        # operand lane group k holds pixel x = 2 r + (k & 1), y = 2 m + (k >> 1) (mm_basis)
        assert xs == [2 * r + (k & 1) for k in range(4)] and ys == [2 * m + (k >> 1) for k in range(4)]
        # the skip mask tests bit 16 m + 2 r of (hit | hit >> 1 | (hit | hit >> 1) >> 8): the block's four pixels
        bits = {16 * m + 2 * r + d for d in (0, 1, 8, 9)}
        assert bits == {y * 8 + x for x, y in zip(xs, ys)}


def test_shifted_quadrant_moments_equal_the_reference_sums():
    rng = np.random.default_rng(0)
    W, H = 640.0, 360.0
    ddelx, ddely = 0.5 * W, 0.5 * H
    x0, y0 = 48.0, 120.0                                   # quadrant origin
    lx, ly = np.meshgrid(np.arange(8.0), np.arange(8.0))   # lane = 8 y + x
    cx, cy = (lx - 3.5).ravel(), (ly - 3.5).ravel()
    px, py = x0 + lx.ravel(), y0 + ly.ravel()
    for trial in range(200):
        far = trial % 3 == 0
        X = x0 + 3.5 + rng.uniform(-400, 400) if far else x0 + rng.uniform(-6, 14)
        Y = y0 + 3.5 + rng.uniform(-400, 400) if far else y0 + rng.uniform(-6, 14)
        A, B, C, O = rng.uniform(0.01, 3), rng.uniform(-1, 1), rng.uniform(0.01, 3), rng.uniform(0.05, 1)
        q = rng.standard_normal(64) * (rng.uniform(size=64) < 0.4)     # G * dL_dalpha, zero where the entry does not hit
        u = rng.uniform(0, 1, 64) * (q != 0)                           # alpha * T
        dpx = rng.uniform(-1, 1, (3, 64))
        # the reference, pixel by pixel (dx = mean - pixel; backward.cu:487-531)
        dx, dy = X - px, Y - py
        dL_dG = O * q / 1.0                                            # q already carries G: dL_dG * G = O * q
        ref = dict(mean_x=np.sum(-dL_dG * (dx * A + dy * B)) * ddelx, mean_y=np.sum(-dL_dG * (dy * C + dx * B)) * ddely,
                   conic_x=np.sum(-0.5 * dL_dG * dx * dx), conic_y=np.sum(-0.5 * dL_dG * dx * dy),
                   conic_w=np.sum(-0.5 * dL_dG * dy * dy), opacity=np.sum(q), colour=dpx @ u)
        # the kernel: moments about the quadrant centre (what the MFMAs deliver), then the per-entry shift
        S1, Sx, Sy = q.sum(), (q * cx).sum(), (q * cy).sum()
        Sxx, Sxy, Syy = (q * cx * cx).sum(), (q * cx * cy).sum(), (q * cy * cy).sum()
        bx, by = X - (x0 + 3.5), Y - (y0 + 3.5)
        Dx, Dy = bx * S1 - Sx, by * S1 - Sy
        got = dict(mean_x=(O * -ddelx) * (A * Dx + B * Dy), mean_y=(O * -ddely) * (B * Dx + C * Dy),
                   conic_x=-0.5 * O * (bx * Dx - bx * Sx + Sxx), conic_y=-0.5 * O * (by * Dx - bx * Sy + Sxy),
                   conic_w=-0.5 * O * (by * Dy - by * Sy + Syy), opacity=S1, colour=dpx @ u)
        for k in ref:
            scale = np.abs(q).sum() * (1 + bx * bx + by * by) * (ddelx if k.startswith("mean") else 1.0) * 3 + 1e-30
            assert np.all(np.abs(np.asarray(got[k]) - np.asarray(ref[k])) <= 1e-12 * scale), (trial, k, got[k], ref[k])


def test_hand_over_addresses_each_component_of_each_entry_once():
    """lane l = 16 g + j owns column j (q of entry j & 7 for j < 8, u for j >= 8) and basis row group g; after the DPP moves
    (row_shr:4 into banks 1, 3; row_shl:4 into banks 0, 2; row_ror:8 in row 3) instruction A serves entries 0..3 and B 4..7."""
    seen = {"A": [], "B": []}
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        c2 = 8 if g == 2 else g
        on_a = j < 8 if g < 3 else (j < 4 or j >= 8)
        on_b = j < 8 if g < 3 else j >= 4
        c_a = (2 + g if j < 4 else c2) if g < 3 else (6 if j >= 12 else 5 if j >= 8 else 7)
        c_b = (c2 if j < 4 else 2 + g) if g < 3 else (5 if j >= 12 else 6 if j >= 8 else 7)
        if on_a:
            seen["A"].append((j & 3, c_a))
        if on_b:
            seen["B"].append((j & 3, c_b))
        # where the value comes from: own o1 / o2 of the lane four below or above / o3 of lane + 8 in the same row
        if g < 3 and j < 8:
            own_entry = j & 7
            assert (own_entry < 4) == (j < 4)
    want = sorted((e, c) for e in range(4) for c in range(9))
    assert sorted(seen["A"]) == want and sorted(seen["B"]) == want
