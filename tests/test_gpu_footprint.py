"""GPU tests (-m gpu) of footprint clipping, the product's DEFAULT list construction (include/gsr.h, gsr_params.reference_lists = 0;
csrc/tile_cull.hpp clip_rect_to_footprint): a Gaussian emits (tile, Gaussian) pairs only for the tiles of the reference's
rectangle (CR/auxiliary.h:46-56) in which alpha can reach 1/255.  The other parity modules compare the PRIVATE lists with the
reference's element by element and therefore run with reference_lists = 1; this module holds the default mode against that
one and against the reference build / oracle on every scene, on random cases and at the benchmark's full size:
  * out_color, final_T, radii, tiles_touched, num_rendered, per-Gaussian floats: bit-identical;
  * lists: the reference's lists with entries removed, each removed pair dead at every pixel of its tile (float32 replay);
  * n_contrib: the same Gaussian through the shorter lists;
  * gradients: inside the usual bars against the oracle.
"""
import os

import numpy as np
import pytest

import util
from util import SCENES, build_scene, run_product, seeded_dL, check_clipped_equivalent

pytestmark = pytest.mark.gpu


def _ref():
    return util.reference_build("strict")


@pytest.mark.parametrize("name", SCENES)
def test_clipped_lists_render_what_the_full_lists_render(name, oracle, gpu_device):
    s = build_scene(name)
    dL = seeded_dL(s)
    a, ga = run_product(s, gpu_device, dL_dpix=dL, reference_lists=True)
    b, gb = run_product(s, gpu_device, dL_dpix=dL, reference_lists=False)
    kept, total = check_clipped_equivalent(a, b, name)
    if s.P == 0:
        return
    _, go = oracle.forward_backward(s, dL)
    util.check_grads(gb, go, name + " (clipped lists) vs oracle")
    print("%s: %d of %d pairs kept (%.0f %%)" % (name, kept, total, 100.0 * kept / max(total, 1)))


def test_clipping_removes_a_substantial_share_of_large_splats_pairs(gpu_device):
    """Guards against the clipping being silently off: splats of ~40 tiles at opacity ~0.3 lose a third of their pairs (the
    benchmark cloud at 1080p: 32 %, test_fullsize_clipped_bit_exact_vs_reference_build prints it)."""
    s = build_scene("big_splats")
    b, _ = run_product(s, gpu_device, reference_lists=False)
    assert 0 < b["L"] < 0.7 * b["R"], (b["L"], b["R"])


def test_public_api_default_is_clipped_and_matches_the_reference_build(gpu_device):
    """Through GaussianRasterizer (no test switch): the default call renders the reference build's image bit for bit."""
    import torch
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _native as N
    ref = _ref()
    s = build_scene("capsule_circle")
    r = ref.forward(s)
    dev = gpu_device
    assert N._REFERENCE_LISTS[0] is False, "the product's default must be footprint clipping"
    st = GaussianRasterizationSettings(
        image_height=s.H, image_width=s.W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=torch.from_numpy(s.bg).to(dev),
        scale_modifier=1.0, viewmatrix=torch.from_numpy(s.viewmatrix.reshape(4, 4)).to(dev),
        projmatrix=torch.from_numpy(s.projmatrix.reshape(4, 4)).to(dev), sh_degree=s.sh_degree,
        campos=torch.from_numpy(s.campos).to(dev), prefiltered=False, debug=False)
    t = lambda x: torch.from_numpy(x).to(dev)  # noqa: E731
    img, radii = GaussianRasterizer(st)(means3D=t(s.means3D), means2D=torch.zeros_like(t(s.means3D)), shs=t(s.shs),
                                        opacities=t(s.opacities.reshape(-1, 1)), scales=t(s.scales), rotations=t(s.rotations))
    assert img.cpu().numpy().tobytes() == r["out_color"].tobytes()
    assert np.array_equal(radii.cpu().numpy(), r["radii"])


N_FUZZ = int(os.environ.get("GSR_FOOTPRINT_FUZZ_CASES", "96"))   # (a 6 000-case run: profiles/r04_footprint_clipping.txt)


@pytest.mark.parametrize("i", range(N_FUZZ))
def test_random_case_clipped_vs_reference_build(i, gpu_device):
    from test_gpu_fuzz import _case
    ref = _ref()
    s, _ = _case(i)
    r = ref.forward(s)
    a, _ = run_product(s, gpu_device, reference_lists=True)
    b, _ = run_product(s, gpu_device, reference_lists=False)
    assert b["out_color"].tobytes() == r["out_color"].tobytes(), "case %d" % i
    assert b["R"] == r["R"] and np.array_equal(b["radii"], r["radii"])
    check_clipped_equivalent(a, b, "fuzz case %d" % i)


@pytest.mark.parametrize("name", ["thuman256_1080p", "thuman800k_1080p", "mesh2m_4k"])
def test_fullsize_clipped_bit_exact_vs_reference_build(name, gpu_device):
    from test_gpu_fullsize import _scene, CONFIGS
    ref = _ref()
    s = _scene(CONFIGS[name], voxel_exact=(name == "thuman256_1080p"))
    r = ref.forward(s)
    b, _ = run_product(s, gpu_device, reference_lists=False, light=True)
    assert b["R"] == r["R"]
    np.testing.assert_array_equal(b["radii"], r["radii"])
    assert b["out_color"].tobytes() == r["out_color"].tobytes()
    print("%s: %d of %d pairs kept (%.0f %%)" % (name, b["L"], b["R"], 100.0 * b["L"] / b["R"]))


def test_fullsize_clipped_lists_are_sublists_and_gradients_hold(gpu_device):
    from test_gpu_fullsize import _scene, CONFIGS
    ref = _ref()
    s = _scene(CONFIGS["thuman800k_1080p"])
    dL = seeded_dL(s)
    _, gr = ref.forward_backward(s, dL)
    a, _ = run_product(s, gpu_device, reference_lists=True)
    b, gb = run_product(s, gpu_device, dL_dpix=dL, reference_lists=False)
    check_clipped_equivalent(a, b, "800K/1080p", check_dead=False)    # (the dead-pair replay of 4 M pairs x 256 pixels is left to the small scenes)
    util.check_grads(gb, gr, "800K/1080p clipped vs reference build")
