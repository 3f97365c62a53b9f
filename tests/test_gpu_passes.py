"""GPU tests of the caller-side glue: colour-only re-render on shared geometry is bit-identical to a
full forward, and the fused four-pass renderer equals four literal `_rasterize`-style calls."""
import numpy as np
import pytest
import torch

import util
from util import build_scene, run_product

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("name", ["capsule_circle", "big_splats", "culled_mix"])
def test_recolor_is_bit_identical_to_a_full_forward(name, gpu_device):
    from diff_gaussian_rasterization import _native as N
    from oracle.oracle import Scene
    dev = gpu_device
    s = build_scene(name)
    e = torch.empty(0)
    args = (_t(s.bg, dev), _t(s.means3D, dev), e, _t(s.opacities, dev), _t(s.scales, dev), _t(s.rotations, dev), 1.0, e,
            _t(s.viewmatrix.reshape(4, 4), dev), _t(s.projmatrix.reshape(4, 4), dev), s.tanfovx, s.tanfovy, s.H, s.W,
            _t(s.shs, dev), s.sh_degree, _t(s.campos, dev), False, False)
    R, rgb, radii, geom, binning, img = N.rasterize_gaussians(*args, need_backward=False)
    rng = np.random.default_rng(5)
    for colors in (s.means3D, np.ones_like(s.means3D), rng.uniform(-1, 1, s.means3D.shape).astype(np.float32)):
        got = N.recolor(args[0], args[1], _t(colors, dev), e, 0, args[16], s.H, s.W, R, geom, binning, img).cpu().numpy()
        s2 = Scene(W=s.W, H=s.H, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=s.bg, means3D=s.means3D, opacities=s.opacities,
                   viewmatrix=s.viewmatrix, projmatrix=s.projmatrix, campos=s.campos, colors_precomp=colors, scales=s.scales,
                   rotations=s.rotations)
        want = run_product(s2, dev)[0]["out_color"]
        assert got.tobytes() == want.tobytes()
    # and back to the SH colours: identical to the original forward
    again = N.recolor(args[0], args[1], e, args[14], s.sh_degree, args[16], s.H, s.W, R, geom, binning, img)
    assert torch.equal(again, rgb)


def test_render_passes_equals_four_literal_passes(gpu_device):
    from pcrender import raster_passes as rp, camera, synth
    dev = gpu_device
    cloud = synth.make_cloud("synth-THuman-256", seed=0, P=30000)
    g = synth.make_gaussians(cloud, profile="inference", seed=1)
    sf = cloud["scale_factor"]
    radius = np.sqrt(3) / sf * 6
    means, shs = _t(g["means3D"], dev), _t(g["shs"], dev)
    opac, rots = _t(g["opacities"], dev), _t(g["rotations"], dev)
    decoded_s = _t((g["scales"] / radius).astype(np.float32), dev)           # what the predictor emits; the caller multiplies by radius
    normals = torch.nn.functional.normalize(means, dim=-1)
    Hs = camera.circle_path(4, 0, 3, [90, 0])
    h = w = 96
    bg = torch.ones(3)
    fused = rp.render_passes(means, opac, decoded_s, rots, shs, Hs, h, w, 45.0, bg, sf, normals=normals, sh_degree=1,
                             super_sample_rate=2)
    Hb = Hs.unsqueeze(0)
    with torch.no_grad():
        lit = dict(
            rgb=rp.rasterize_views([means], [opac], [decoded_s], [rots], Hb, h, w, 45.0, bg, sf, shs_list=[shs], super_sample_rate=2),
            xyz_w=rp.rasterize_views([means], [opac], [decoded_s], [rots], Hb, h, w, 45.0, bg, sf, colors_list=[means], super_sample_rate=2),
            hitmap=rp.rasterize_views([means], [opac], [decoded_s], [rots], Hb, h, w, 45.0, bg, sf, colors_list=[torch.ones_like(means)], super_sample_rate=2),
            normal=rp.rasterize_views([means], [opac], [decoded_s], [rots], Hb, h, w, 45.0, bg, sf, colors_list=[normals], super_sample_rate=2,
                                      normalize_camera_normal=True))
    for k in ("rgb", "xyz_w", "hitmap", "normal"):
        assert fused[k].shape == (1, 4, h, w, 3) == lit[k].shape
        assert torch.equal(fused[k], lit[k]), k
    # the literal pattern with the views of a batch item submitted together
    with torch.no_grad():
        tog = rp.rasterize_views([means], [opac], [decoded_s], [rots], Hb, h, w, 45.0, bg, sf, shs_list=[shs], super_sample_rate=2,
                                 batch_views=True)
    assert torch.equal(tog, lit["rgb"])
    # hit map: 1 where the body covers the pixel (background 1 too here), strictly inside [0, 1]
    assert float(fused["hitmap"].min()) > 0.99 and float(fused["hitmap"].max()) <= 1.0 + 1e-6


def test_supersample_downfilter_is_the_references_bilinear(gpu_device):
    """super_sample_rate = 2 with align_corners=False bilinear to half size is the 2x2 box mean."""
    from pcrender import raster_passes as rp
    x = torch.rand(2, 3, 8, 12, device=gpu_device)
    y = rp._finish([x[0], x[1]], 1, 2, 4, 6, 2)
    want = x.reshape(2, 3, 4, 2, 6, 2).mean(dim=(3, 5)).reshape(1, 2, 3, 4, 6).permute(0, 1, 3, 4, 2)
    torch.testing.assert_close(y, want, atol=1e-6, rtol=0)


def test_render_passes_equal_the_replay_of_the_reference_callers_trace(gpu_device):
    """tests/golden/py_rasterize_calls.npz holds every rasterizer call PCML_Render.render / Simple_Render.render of the reference
    issue on a toy input (tests/test_cpu_call_trace.py pins the glue to it on the CPU).  Here the recorded calls are REPLAYED through
    the product -- GaussianRasterizer(settings)(arguments) exactly as recorded, then the reference's stack / down-filter / permute --
    and the fused render_passes (one submission, one render for the four passes) has to return the same images bit for bit."""
    import os
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from pcrender import raster_passes as rp
    dev = gpu_device
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "py_rasterize_calls.npz"))
    n, h, w, ss, fov, sf, offset = (float(x) for x in G["args"])
    h, w, ss, offset = int(h), int(w), int(ss), int(offset)
    H = torch.from_numpy(G["H_c2w"])[0]
    q = H.shape[0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731

    def replay(k):
        g = lambda a: (t(G["call%02d_%s" % (k, a)]) if "call%02d_%s" % (k, a) in G.files else None)   # noqa: E731
        sc = G["call%02d_scalars" % k]
        st = GaussianRasterizationSettings(
            image_height=int(sc[0]), image_width=int(sc[1]), tanfovx=float(sc[2]), tanfovy=float(sc[3]), bg=g("bg"),
            scale_modifier=float(sc[4]), viewmatrix=g("viewmatrix"), projmatrix=g("projmatrix"), sh_degree=int(sc[5]), campos=g("campos"),
            prefiltered=bool(sc[6]), debug=bool(sc[7]))
        with torch.no_grad():
            img, _ = GaussianRasterizer(st)(means3D=g("means3D"), means2D=g("means2D"), shs=g("shs"), colors_precomp=g("colors_precomp"),
                                            opacities=g("opacities"), scales=g("scales"), rotations=g("rotations"), cov3D_precomp=None)
        return img

    n_pcml, n_simple = (int(x) for x in G["n_calls"])
    # PCML_Render.render: calls 0 .. 4q-1 in the order xyz_w, rgb, hitmap, normal
    want = {name: rp._finish([replay(i * q + j) for j in range(q)], 1, q, h, w, ss)
            for i, name in enumerate(("xyz_w", "rgb", "hitmap", "normal"))}
    means = rp.pcgc_rescale(t(G["pcml_decoded_primitives"]).float(), offset, sf)
    fused = rp.render_passes(means, t(G["pcml_decoded_o"]), t(G["pcml_decoded_s"]), t(G["pcml_decoded_r"]), t(G["pcml_decoded_sh"]), H,
                             h, w, fov, torch.ones(3), sf, normals=t(G["pcml_decoded_n"]), sh_degree=1, super_sample_rate=ss)
    for name in want:
        assert torch.equal(fused[name], want[name]), name
    assert float((want["rgb"] - 1.0).abs().max()) > 0.05      # the toy cloud is on screen: the images are not pure background
    # Simple_Render.render: rgb, xyz_w, hitmap (scales as they are, opacity 1) through the literal path's simple mode
    # (the primitives are formed on the CPU like the fixture's: torch divides by a Python scalar as a multiplication by its
    # reciprocal on the GPU, one ulp away from the CPU's division in RGB2SH -- torch's arithmetic, not the glue's)
    prim = {k: v.to(dev) for k, v in rp.simple_primitives(torch.from_numpy(G["simple_xyz"]), torch.from_numpy(G["simple_rgb"]), sigma=1.5,
                                                          scale_factor=sf, voxelized=True, offset=offset).items()}
    want2 = {name: rp._finish([replay(n_pcml + i * q + j) for j in range(q)], 1, q, h, w, ss)
             for i, name in enumerate(("rgb", "xyz_w", "hitmap"))}
    with torch.no_grad():
        lit = rp.literal_passes(prim["means3D"], prim["opacities"], prim["scales"], prim["rotations"], prim["shs"], H, h, w, fov, 0.0, sf,
                                sh_degree=1, super_sample_rate=ss, simple=True)
    for name in want2:
        assert torch.equal(lit[name], want2[name]), name
