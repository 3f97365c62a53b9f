"""GPU tests (-m gpu) of the view-batched entry points (C ABI gsr_forward_batch / gsr_backward_batch) and of the
capacity-based binning arena (no host round trip inside a frame; GSR_RETRY when the arena is too small):

  * a batch of V views against V oracle forwards / backwards (gradients: the sum over views);
  * bit-identical images / radii / pair counts between a batch and V single-view calls;
  * the retry path: a forward started with a far too small arena still returns the right image;
  * the autograd extension rasterize_views against V GaussianRasterizer calls.
"""
import numpy as np
import pytest
import torch

import util
from util import check_grads

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _views_scene(n_views=3, P=12000, W=208, H=176, profile="training"):
    """One cloud, n_views circle cameras (view 0 is axis aligned: depth ties on the voxelised cloud)."""
    from pcrender import camera, synth
    cloud = synth.make_cloud("synth-THuman-256", seed=0, P=P)
    g = synth.make_gaussians(cloud, profile=profile, seed=1)
    views = camera.circle_views(12, fov_deg=45.0, width_px=W, height_px=H)
    pick = [0, 1, 5, 7, 10][:n_views]
    return g, [views[i] for i in pick], W, H


def _batch_args(g, views, W, H, dev, bg=(1, 1, 1)):
    e = torch.empty(0)
    vm = torch.stack([v["viewmatrix"] for v in views]).to(dev)
    pm = torch.stack([v["projmatrix"] for v in views]).to(dev)
    cp = torch.stack([v["campos"] for v in views]).to(dev)
    return (_t(np.asarray(bg, np.float32), dev), _t(g["means3D"], dev), e, _t(g["opacities"], dev), _t(g["scales"], dev),
            _t(g["rotations"], dev), 1.0, e, vm, pm, views[0]["tanfovx"], views[0]["tanfovy"], H, W, _t(g["shs"], dev),
            g["sh_degree"], cp, False, False)


def test_batch_forward_backward_vs_oracle(oracle, gpu_device):
    from diff_gaussian_rasterization import _native as N
    dev = gpu_device
    g, views, W, H = _views_scene(3)
    args = _batch_args(g, views, W, H, dev)
    counts, color, radii, geom, binning, img = N.rasterize_gaussians_batch(*args, need_backward=True)
    V = len(views)
    assert color.shape == (V, 3, H, W) and radii.shape == (V, g["means3D"].shape[0])
    rng = np.random.default_rng(9)
    dL = rng.uniform(-1, 1, (V, 3, H, W)).astype(np.float32)
    grads = N.rasterize_gaussians_backward_batch(args[0], args[1], radii, args[2], args[4], args[5], 1.0, args[7], args[8], args[9],
                                                 args[10], args[11], _t(dL, dev), args[14], args[15], args[16], geom, binning, img, False)
    names = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot")
    gp = {n: x.cpu().numpy() for n, x in zip(names, grads)}
    total = None
    flips = False
    for v, view in enumerate(views):
        s = util.scene_from(g, view, W, H, bg=(1, 1, 1))
        o, go = oracle.forward_backward(s, dL[v])
        assert counts[v] == o["R"]
        np.testing.assert_array_equal(radii[v].cpu().numpy(), o["radii"])
        err = np.abs(color[v].cpu().numpy() - o["out_color"]).max(axis=0)
        assert (err > 1e-4).mean() <= 2e-3
        flips = flips or bool((err > 1e-4).any())
        go["dL_dopacity"] = go["dL_dopacity"].reshape(-1, 1)
        total = go if total is None else {k: total[k] + go[k] for k in total}
    if not flips:
        check_grads(gp, total, "batch of %d views vs summed oracle gradients" % V)


def test_batch_equals_single_view_calls_bit_for_bit(gpu_device):
    from diff_gaussian_rasterization import _native as N
    dev = gpu_device
    g, views, W, H = _views_scene(5, P=20000, W=320, H=240)
    args = _batch_args(g, views, W, H, dev)
    counts, color, radii, geom, binning, img = N.rasterize_gaussians_batch(*args, need_backward=False)
    P, V = g["means3D"].shape[0], len(views)
    for v, view in enumerate(views):
        one = list(args)
        one[8], one[9], one[16] = args[8][v], args[9][v], args[16][v]
        R, c1, r1, g1, b1, i1 = N.rasterize_gaussians(*one, need_backward=False)
        assert R == counts[v]
        assert torch.equal(c1, color[v]) and torch.equal(r1, radii[v])
        # the private arrays of view v inside the batch arenas: sorted lists, ranges, per-pixel bookkeeping
        for name in ("POINT_LIST", "POINT_LIST_KEYS", "RANGES", "N_CONTRIB", "FINAL_T", "TILES_TOUCHED"):
            a = N.query(name, P, W, H, R, geom, binning, img, view=v, n_views=V)
            b = N.query(name, P, W, H, R, g1, b1, i1)
            assert torch.equal(a, b), (v, name)


def test_too_small_arena_is_retried_transparently(gpu_device):
    from diff_gaussian_rasterization import _native as N
    dev = gpu_device
    g, views, W, H = _views_scene(2, P=15000)
    args = _batch_args(g, views, W, H, dev)
    counts_ok, color_ok, radii_ok, *_ = N.rasterize_gaussians_batch(*args, need_backward=False)
    assert min(counts_ok) > 20000
    # 1000 pairs per view: every view overflows; the C ABI reports GSR_RETRY and the binding repeats the binning half
    counts, color, radii, geom, binning, img = N.rasterize_gaussians_batch(*args, need_backward=True, capacity=1000)
    assert counts == counts_ok and torch.equal(color, color_ok) and torch.equal(radii, radii_ok)
    # the raw status code, and the arenas of a retried forward serve the backward
    p, keep = N._params(*args, need_backward=False)
    import ctypes as C
    small = torch.empty((2 * N.lib.gsr_binning_bytes(1000),), dtype=torch.uint8, device=dev)
    cnt = (C.c_int64 * 2)()
    out = torch.empty_like(color)
    rad = torch.empty_like(radii)
    g2 = torch.empty_like(geom)
    i2 = torch.empty_like(img)
    with torch.cuda.device(dev):
        rc = N.lib.gsr_forward_batch(C.byref(p), 2, g2.data_ptr(), g2.numel(), i2.data_ptr(), i2.numel(), small.data_ptr(), small.numel(),
                                     rad.data_ptr(), out.data_ptr(), cnt, 0, torch.cuda.current_stream(dev).cuda_stream)
    assert rc == N.GSR_RETRY and list(cnt) == counts_ok and b"resume" in N.lib.gsr_last_error()
    dL = torch.ones_like(color)
    grads = N.rasterize_gaussians_backward_batch(args[0], args[1], radii, args[2], args[4], args[5], 1.0, args[7], args[8], args[9],
                                                 args[10], args[11], dL, args[14], args[15], args[16], geom, binning, img, False)
    assert all(torch.isfinite(x).all() for x in grads) and float(grads[3].abs().max()) > 0


def test_rasterize_views_autograd_vs_per_view_calls(gpu_device):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, rasterize_views
    dev = gpu_device
    g, views, W, H = _views_scene(4, P=10000, W=160, H=144)
    bg = torch.ones(3, device=dev)
    sts = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=v["tanfovx"], tanfovy=v["tanfovy"], bg=bg,
                                         scale_modifier=1.0, viewmatrix=v["viewmatrix"].to(dev), projmatrix=v["projmatrix"].to(dev),
                                         sh_degree=g["sh_degree"], campos=v["campos"].to(dev), prefiltered=False, debug=False)
           for v in views]
    G = torch.from_numpy(np.random.default_rng(3).uniform(-1, 1, (len(views), 3, H, W)).astype(np.float32)).to(dev)

    def leaves():
        L = {k: _t(g[k], dev).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
        L["means2D"] = torch.zeros_like(L["means3D"], requires_grad=True)
        return L

    A = leaves()
    imgs, radii = rasterize_views(A["means3D"], A["means2D"], A["opacities"], sts, shs=A["shs"], scales=A["scales"], rotations=A["rotations"])
    (imgs * G).sum().backward()
    B = leaves()
    loss = 0
    for v, st in enumerate(sts):
        im, rd = GaussianRasterizer(st)(means3D=B["means3D"], means2D=B["means2D"], shs=B["shs"], opacities=B["opacities"],
                                        scales=B["scales"], rotations=B["rotations"])
        assert torch.equal(im, imgs[v]) and torch.equal(rd, radii[v])
        loss = loss + (im * G[v]).sum()
    loss.backward()
    gp = dict(dL_dmean2D=A["means2D"].grad, dL_dopacity=A["opacities"].grad, dL_dmean3D=A["means3D"].grad, dL_dsh=A["shs"].grad,
              dL_dscale=A["scales"].grad, dL_drot=A["rotations"].grad)
    go = dict(dL_dmean2D=B["means2D"].grad, dL_dopacity=B["opacities"].grad, dL_dmean3D=B["means3D"].grad, dL_dsh=B["shs"].grad,
              dL_dscale=B["scales"].grad, dL_drot=B["rotations"].grad)
    check_grads({k: x.cpu().numpy() for k, x in gp.items()}, {k: x.cpu().numpy() for k, x in go.items()}, "rasterize_views",
                names=tuple(gp))


def test_no_host_sync_between_kernels_of_a_frame(gpu_device):
    """The frame is enqueued without waiting for the device: with the stream blocked behind a long-running kernel, the
    forward call must still return only after its own final wait, and a second frame's capacity comes from the first's."""
    from diff_gaussian_rasterization import _native as N
    dev = gpu_device
    g, views, W, H = _views_scene(1, P=8000)
    args = _batch_args(g, views, W, H, dev)
    N.reset_capacity_hints()
    c1, img1, *_ = N.rasterize_gaussians_batch(*args, need_backward=False)      # first frame: count, then bind
    key = N._cap_key(dev, 8000, W, H)
    # the arena holds the LISTS: their pair count (gsr_last_list_pairs; footprint clipping makes it smaller than num_rendered)
    import ctypes as C
    pairs = (C.c_int64 * 1)()
    assert N.lib.gsr_last_list_pairs(pairs, 1) == 0
    assert 0 < N._CAP_HINT[key] == pairs[0] <= c1[0]
    c2, img2, _, _, binning, _ = N.rasterize_gaussians_batch(*args, need_backward=False)   # second: one submission
    assert c2 == c1 and torch.equal(img1, img2)
    assert binning.numel() >= N.lib.gsr_binning_bytes(int(pairs[0] * N.CAP_SLACK))


def test_batch_edge_cases_vs_oracle(oracle, gpu_device):
    """Small clouds with many views (views spread over grid.y in the preprocess kernel), a view that sees nothing
    (num_rendered = 0: pure background), an empty cloud, and the maximum view count check."""
    from diff_gaussian_rasterization import _native as N
    from pcrender import camera, synth
    dev = gpu_device
    W, H = 72, 56
    g = synth.random_scene(60, W, H, seed=3, sh_degree=2, spread=0.8, scale=0.1)
    g["means3D"][:, 2] -= 3.0                                   # around the origin, like the circle cameras expect
    views = camera.circle_views(12, fov_deg=45.0, width_px=W, height_px=H)
    far = dict(views[0])
    far["viewmatrix"] = views[0]["viewmatrix"].clone()
    far["viewmatrix"][3, 2] = -50.0                             # camera-space z = z - 50: everything behind the near plane
    vs = views[:9] + [far] + views[9:] + views[:4]              # 17 views
    args = _batch_args(g, vs, W, H, dev, bg=(0.2, 0.5, 0.7))
    counts, color, radii, geom, binning, img = N.rasterize_gaussians_batch(*args, need_backward=True)
    assert counts[9] == 0 and int(radii[9].abs().max()) == 0
    bgimg = torch.tensor([0.2, 0.5, 0.7], device=dev).view(3, 1, 1).expand(3, H, W)
    assert torch.allclose(color[9], bgimg)
    for v in (0, 5, 9, 16):
        s = util.scene_from(g, vs[v], W, H, bg=(0.2, 0.5, 0.7))
        o = oracle.forward(s)
        assert counts[v] == o["R"]
        np.testing.assert_array_equal(radii[v].cpu().numpy(), o["radii"])
        err = np.abs(color[v].cpu().numpy() - o["out_color"]).max()
        assert err <= 1e-4, (v, err)
    dL = torch.ones_like(color)
    grads = N.rasterize_gaussians_backward_batch(args[0], args[1], radii, args[2], args[4], args[5], 1.0, args[7], args[8], args[9],
                                                 args[10], args[11], dL, args[14], args[15], args[16], geom, binning, img, False)
    assert all(torch.isfinite(x).all() for x in grads)
    # empty cloud: zero images (not the background), like the reference's P == 0 shortcut
    e = torch.empty(0)
    z = torch.zeros((0, 3), device=dev)
    cnt0, col0, rad0, *_ = N.rasterize_gaussians_batch(args[0], z, e, torch.zeros((0, 1), device=dev), z, torch.zeros((0, 4), device=dev),
                                                      1.0, e, args[8][:3], args[9][:3], args[10], args[11], H, W,
                                                      torch.zeros((0, 9, 3), device=dev), 2, args[16][:3], False, False)
    assert cnt0 == [0, 0, 0] and not col0.any() and rad0.shape == (3, 0)
    with pytest.raises(RuntimeError, match="view count"):
        N.rasterize_gaussians_batch(*(args[:8] + (args[8][:1].expand(300, 4, 4), args[9][:1].expand(300, 4, 4)) + args[10:16] +
                                      (args[16][:1].expand(300, 3),) + args[17:]))


def test_batch_debug_and_prefiltered_trap(gpu_device):
    """debug = True synchronises and checks after every launch of a batch (the reference's CHECK_CUDA); prefiltered = True with
    a Gaussian behind some view's near plane raises with the reference's text instead of trapping the device."""
    from diff_gaussian_rasterization import _native as N
    dev = gpu_device
    g, views, W, H = _views_scene(3, P=3000)
    args = list(_batch_args(g, views, W, H, dev))
    ok_counts, ok_color, *_ = N.rasterize_gaussians_batch(*args, need_backward=False)
    dbg = list(args)
    dbg[18] = True
    c2, col2, *_ = N.rasterize_gaussians_batch(*dbg, need_backward=False)
    assert c2 == ok_counts and torch.equal(col2, ok_color)
    pre = list(args)
    pre[17] = True
    c3, col3, *_ = N.rasterize_gaussians_batch(*pre, need_backward=False)          # nothing is culled: fine
    assert c3 == ok_counts and torch.equal(col3, ok_color)
    m = pre[1].clone()
    m[7] = torch.tensor([0.0, 0.0, 50.0], device=dev)                                # far behind every camera looking at the origin
    pre[1] = m
    # the point lies behind the camera of at least one of the circle views
    with pytest.raises(RuntimeError, match="Point is filtered although prefiltered is set"):
        N.rasterize_gaussians_batch(*pre, need_backward=False)


@pytest.mark.parametrize("nx,V,need_backward", [(8, 3, False), (4, 1, False), (8, 2, True)])
def test_extra_channels_equal_separate_colour_passes(gpu_device, nx, V, need_backward):
    """gsr_forward_batch_channels: every extra channel is bit-for-bit what a full call with that channel as colors_precomp
    renders (same alphas, transmittances and stops), the colour output and the bookkeeping are those of the plain call."""
    from diff_gaussian_rasterization import _native as N
    dev = gpu_device
    g, views, W, H = _views_scene(V)
    args = _batch_args(g, views, W, H, dev, bg=(0.25, 0.25, 0.25))
    P = g["means3D"].shape[0]
    rng = np.random.default_rng(31)
    extra = _t(rng.normal(0, 1, (P, nx)).astype(np.float32), dev)
    scale = _t(rng.choice([-1.0, 1.0, 0.5], (V, nx)).astype(np.float32), dev)
    bgx = _t(rng.uniform(0, 1, nx).astype(np.float32), dev)
    N.reset_capacity_hints()       # first call with a guessed capacity: covers the retry path of the channels entry point too
    r = N.rasterize_gaussians_batch(*args, need_backward=need_backward, extra=(extra, scale, bgx))
    counts, color, radii, geom, binning, img, out_x = r
    assert out_x.shape == (V, nx, H, W)
    c0, color0, radii0, geom0, binning0, img0 = N.rasterize_gaussians_batch(*args, need_backward=need_backward)
    assert counts == c0 and torch.equal(color, color0) and torch.equal(radii, radii0)
    for name in ("FINAL_T", "N_CONTRIB"):
        for v in range(V):
            assert torch.equal(N.query(name, P, W, H, counts[v], geom, binning, img, view=v, n_views=V),
                               N.query(name, P, W, H, c0[v], geom0, binning0, img0, view=v, n_views=V))
    e = torch.empty(0)
    for k0 in range(0, nx, 3):
        ks = [min(k0 + i, nx - 1) for i in range(3)]
        for v in range(V):
            cols = (extra[:, ks] * scale[v, ks]).contiguous()
            a = list(args)
            a[0] = bgx[ks].contiguous()
            a[2], a[14] = cols, e                           # colors_precomp instead of SHs
            a[8], a[9], a[16] = args[8][v], args[9][v], args[16][v]
            _, ref, _, _, _, _ = N.rasterize_gaussians(*a, need_backward=False)
            assert torch.equal(out_x[v, ks], ref), (k0, v)


def test_split_extra_channels_equal_the_per_view_layout(gpu_device):
    """extra_per_view = 2 (include/gsr.h): channels 0..3 as one [P,4] array shared by the views, channels 4..7 as [V,P,4] -- half the
    memory of a [V,P,8] array when only the last four depend on the view (world xyz + hit value vs the normals turned per view).
    Bit for bit the result of the interleaved per-view layout, and the colour output is the plain call's."""
    from diff_gaussian_rasterization import _native as N
    dev = gpu_device
    V = 3
    g, views, W, H = _views_scene(V)
    args = _batch_args(g, views, W, H, dev, bg=(0.5, 0.5, 0.5))
    P = g["means3D"].shape[0]
    rng = np.random.default_rng(47)
    lo = _t(rng.normal(0, 1, (P, 4)).astype(np.float32), dev)
    hi = _t(rng.normal(0, 1, (V, P, 4)).astype(np.float32), dev)
    bgx = _t(rng.uniform(0, 1, 8).astype(np.float32), dev)
    scale = _t(rng.choice([-1.0, 1.0, 0.5], (V, 8)).astype(np.float32), dev)
    full = torch.cat([lo.unsqueeze(0).expand(V, P, 4), hi], dim=2).contiguous()
    a = N.rasterize_gaussians_batch(*args, need_backward=False, extra=(full, scale, bgx))
    b = N.rasterize_gaussians_batch(*args, need_backward=False, extra=((lo, hi), scale, bgx))
    assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert b[6].shape == (V, 8, H, W) and torch.equal(a[6], b[6])
    with pytest.raises(RuntimeError, match="split extra channels"):
        N.rasterize_gaussians_batch(*args, need_backward=False, extra=((lo, hi[:2]), scale, bgx))


def test_extra_channels_argument_checks(gpu_device):
    from diff_gaussian_rasterization import _native as N
    dev = gpu_device
    g, views, W, H = _views_scene(1, P=500)
    args = _batch_args(g, views, W, H, dev)
    P = g["means3D"].shape[0]
    with pytest.raises(RuntimeError, match="extra channels"):
        N.rasterize_gaussians_batch(*args, need_backward=False, extra=(torch.zeros(P, 5, device=dev), None, torch.zeros(5, device=dev)))
    with pytest.raises(RuntimeError, match="bg_extra"):
        N.rasterize_gaussians_batch(*args, need_backward=False, extra=(torch.zeros(P, 4, device=dev), None, torch.zeros(3, device=dev)))
    with pytest.raises(RuntimeError, match="view_scale"):
        N.rasterize_gaussians_batch(*args, need_backward=False,
                                    extra=(torch.zeros(P, 4, device=dev), torch.zeros(2, 4, device=dev), torch.zeros(4, device=dev)))
