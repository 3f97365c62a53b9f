"""The render backward takes its per-entry sums over a quadrant's 64 pixels on the matrix cores, eight entries (two groups of
four) per batch, with a half-filled batch carried over to the next round of 64 list entries and flushed at the end of a slice
(csrc/render_bwd.hip, "wave reduction on the matrix cores").  These scenes aim at the seams of that scheme: stacks of K
splats over the same tiles with K on both sides of every group / batch / round boundary, every one of them contributing to
every pixel (low opacity, no early termination), and splats whose centre lies far outside the quadrants they cover, where
the shift of the second moments from the quadrant centre to the splat centre multiplies the sums by hundreds of pixels.
Gradients are held to the library's usual bars against the plain-C oracle and, where built, the reference build
(CR/backward.cu:399-557 evaluated per pixel with atomics)."""
import numpy as np
import pytest

import util
from util import run_product, check_grads

pytestmark = pytest.mark.gpu

W, H = 80, 48
Z0 = 2.5
COUNTS = [1, 2, 3, 4, 5, 7, 8, 9, 11, 12, 13, 16, 17, 31, 32, 33, 63, 64, 65, 67, 68, 69, 127, 128, 129, 133]


def _stack(K, seed, far=False):
    """K splats over the middle of the image at slightly different depths; far=True puts every other centre 150-400 pixels
    outside the image with a footprint large enough to cover it"""
    rng = np.random.default_rng(seed)
    view = util.identity_camera(W, H, 60.0)
    pm = np.asarray(view["projmatrix"], np.float64).reshape(4, 4)
    kx, ky = 0.5 * W * pm[0, 0], 0.5 * H * pm[1, 1]
    px = rng.uniform(32, 48, K)
    py = rng.uniform(18, 30, K)
    sig = rng.uniform(5.0, 9.0, K)
    if far:
        out = np.arange(K) % 2 == 0
        ang = rng.uniform(0, 2 * np.pi, K)
        dist = rng.uniform(150, 400, K)
        px = np.where(out, W / 2 + dist * np.cos(ang), px)
        py = np.where(out, H / 2 + dist * np.sin(ang), py)
        sig = np.where(out, dist / rng.uniform(1.2, 2.5, K), sig)
    z = Z0 + 0.01 * np.arange(K) + rng.uniform(0, 0.004, K)
    means = np.stack([(px - (W - 1) / 2) / kx * z, (py - (H - 1) / 2) / ky * z, z], 1).astype(np.float32)
    fx = W / (2.0 * view["tanfovx"])
    s_world = (sig * z / fx)[:, None] * rng.uniform(0.6, 1.4, (K, 3))
    q = rng.standard_normal((K, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    g = dict(means3D=means, scales=s_world.astype(np.float32), rotations=q.astype(np.float32),
             opacities=rng.uniform(0.02, 0.12, (K, 1)).astype(np.float32),
             shs=(0.5 * rng.standard_normal((K, 4, 3))).astype(np.float32), sh_degree=1)
    return util.scene_from(g, view, W, H, bg=(0.2, 0.1, 0.3))


def _check(s, tag, oracle, gpu_device, ref):
    dL = util.seeded_dL(s, seed=7)
    fo, go = oracle.forward_backward(s, dL)
    fp, gp = run_product(s, gpu_device, dL_dpix=dL)
    assert (fp["radii"] == fo["radii"]).all() and fp["R"] == fo["R"], tag
    assert (fp["n_contrib"] == fo["n_contrib"]).all(), tag
    check_grads(gp, go, tag + " vs oracle")
    if ref is not None:
        fr, gr = ref.forward_backward(s, dL)
        assert (fp["n_contrib"] == fr["n_contrib"]).all(), tag
        check_grads(gp, gr, tag + " vs reference build")
    return fo


def _ref():
    return util.reference_build("strict")


@pytest.mark.parametrize("K", COUNTS)
def test_stack_sizes_around_group_batch_and_round_boundaries(K, oracle, gpu_device):
    s = _stack(K, seed=100 + K)
    fo = _check(s, "stack of %d" % K, oracle, gpu_device, _ref())
    # the scene does what it is for: somewhere every one of the K splats contributes to a pixel
    assert fo["n_contrib"].max() == K


@pytest.mark.parametrize("K", [6, 9, 40, 70])
def test_centres_far_outside_the_quadrants_they_cover(K, oracle, gpu_device):
    s = _stack(K, seed=300 + K, far=True)
    fo = _check(s, "far stack of %d" % K, oracle, gpu_device, _ref())
    m2 = fo["means2D"]
    off = np.maximum(np.maximum(-m2[:, 0], m2[:, 0] - W), np.maximum(-m2[:, 1], m2[:, 1] - H))
    assert (off > 100).sum() >= K // 2 - 1 and (fo["radii"] > 0).all()
