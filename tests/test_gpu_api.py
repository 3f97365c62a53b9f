"""GPU tests of the drop-in Python surface (GaussianRasterizer / GaussianRasterizationSettings) used the way
the reference caller uses it (simple_raw_render.py:227-288): settings per view, means2D dummy leaf, keyword call,
autograd backward.  Results are checked against the CPU oracle."""
import numpy as np
import pytest
import torch

import util
from util import build_scene, seeded_dL

pytestmark = pytest.mark.gpu


def _settings(s, dev, debug=False):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    # as the reference caller passes them: [1,4,4] non-contiguous views of transposed matrices, campos [1,1,3]
    view = torch.from_numpy(s.viewmatrix.reshape(4, 4).T.copy()).to(dev).t().unsqueeze(0)
    proj = torch.from_numpy(s.projmatrix.reshape(4, 4).T.copy()).to(dev).t().unsqueeze(0)
    assert not view[0].is_contiguous()
    return GaussianRasterizationSettings(
        image_height=s.H, image_width=s.W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=torch.from_numpy(s.bg).to(dev),
        scale_modifier=s.scale_modifier, viewmatrix=view, projmatrix=proj, sh_degree=s.sh_degree,
        campos=torch.from_numpy(s.campos).to(dev).reshape(1, 1, 3), prefiltered=False, debug=debug)


def _leaf(a, dev):
    return torch.from_numpy(a).to(dev).requires_grad_(True)


@pytest.mark.parametrize("name", ["random_aniso", "capsule_circle", "big_splats"])
def test_module_forward_backward_matches_oracle(name, oracle, gpu_device):
    from diff_gaussian_rasterization import GaussianRasterizer
    s = build_scene(name)
    dev = gpu_device
    means3D, shs = _leaf(s.means3D, dev), _leaf(s.shs, dev)
    opac, scales, rots = _leaf(s.opacities.reshape(-1, 1), dev), _leaf(s.scales, dev), _leaf(s.rotations, dev)
    means2D = torch.zeros_like(means3D, dtype=torch.float32, requires_grad=True, device="cuda") + 0
    means2D.retain_grad()
    rasterizer = GaussianRasterizer(_settings(s, dev))
    img, radii = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=None, opacities=opac,
                            scales=scales, rotations=rots, cov3D_precomp=None)
    assert img.shape == (3, s.H, s.W) and radii.shape == (s.P,) and radii.dtype == torch.int32
    dL = seeded_dL(s)
    (img * torch.from_numpy(dL).to(dev)).sum().backward()

    o, go = oracle.forward_backward(s, dL)
    np.testing.assert_array_equal(radii.cpu().numpy(), o["radii"])
    err = np.abs(img.detach().cpu().numpy() - o["out_color"]).max(axis=0)
    assert (err > 1e-4).mean() <= 2e-3
    pairs = [(means3D.grad, "dL_dmean3D"), (means2D.grad, "dL_dmean2D"), (shs.grad, "dL_dsh"), (opac.grad, "dL_dopacity"),
             (scales.grad, "dL_dscale"), (rots.grad, "dL_drot")]
    for g, k in pairs:
        a, b = g.cpu().numpy().reshape(-1).astype(np.float64), go[k].reshape(-1).astype(np.float64)
        assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max() + 1e-30, k


def test_colors_precomp_and_cov3d_paths(oracle, gpu_device):
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = gpu_device
    s = build_scene("colors_precomp")
    colors = _leaf(s.colors_precomp, dev)
    means3D = _leaf(s.means3D, dev)
    means2D = torch.zeros_like(means3D, requires_grad=True)
    img, _ = GaussianRasterizer(_settings(s, dev))(
        means3D=means3D, means2D=means2D, colors_precomp=colors, opacities=torch.from_numpy(s.opacities).to(dev).reshape(-1, 1),
        scales=torch.from_numpy(s.scales).to(dev), rotations=torch.from_numpy(s.rotations).to(dev))
    dL = seeded_dL(s)
    (img * torch.from_numpy(dL).to(dev)).sum().backward()
    o, go = oracle.forward_backward(s, dL)
    assert np.abs(img.detach().cpu().numpy() - o["out_color"]).max() <= 1e-4
    a, b = colors.grad.cpu().numpy().astype(np.float64), go["dL_dcolor"].astype(np.float64)
    assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max()

    s = build_scene("cov3d_precomp")
    cov = _leaf(s.cov3D_precomp, dev)
    means3D = _leaf(s.means3D, dev)
    means2D = torch.zeros_like(means3D, requires_grad=True)
    img, _ = GaussianRasterizer(_settings(s, dev))(
        means3D=means3D, means2D=means2D, shs=torch.from_numpy(s.shs).to(dev), opacities=torch.from_numpy(s.opacities).to(dev).reshape(-1, 1),
        cov3D_precomp=cov)
    dL = seeded_dL(s)
    (img * torch.from_numpy(dL).to(dev)).sum().backward()
    o, go = oracle.forward_backward(s, dL)
    assert np.abs(img.detach().cpu().numpy() - o["out_color"]).max() <= 1e-4
    a, b = cov.grad.cpu().numpy().astype(np.float64), go["dL_dcov3D"].astype(np.float64)
    assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max()


def test_no_grad_inference_call_and_empty_cloud(gpu_device):
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = gpu_device
    s = build_scene("one_gaussian")
    with torch.no_grad():
        means3D = torch.from_numpy(s.means3D).to(dev)
        img, radii = GaussianRasterizer(_settings(s, dev))(
            means3D=means3D, means2D=torch.zeros_like(means3D), shs=torch.from_numpy(s.shs).to(dev),
            opacities=torch.from_numpy(s.opacities).to(dev).reshape(-1, 1), scales=torch.from_numpy(s.scales).to(dev),
            rotations=torch.from_numpy(s.rotations).to(dev))
    assert img.isfinite().all() and int(radii[0]) > 0
    # P == 0: the reference returns an all-zero image, not the background (rasterize_points.cu:68,81)
    e3 = torch.zeros((0, 3), device=dev)
    img, radii = GaussianRasterizer(_settings(s, dev))(
        means3D=e3, means2D=e3, colors_precomp=e3, opacities=torch.zeros((0, 1), device=dev),
        scales=e3, rotations=torch.zeros((0, 4), device=dev))
    assert img.shape == (3, s.H, s.W) and not img.any() and radii.numel() == 0


def test_argument_validation_messages(gpu_device):
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = gpu_device
    s = build_scene("one_gaussian")
    r = GaussianRasterizer(_settings(s, dev))
    m = torch.from_numpy(s.means3D).to(dev)
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        r(means3D=m, means2D=m, opacities=m[:, :1])
    with pytest.raises(Exception, match="Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!"):
        r(means3D=m, means2D=m, opacities=m[:, :1], colors_precomp=m)
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):
        r(means3D=m.reshape(-1), means2D=m, opacities=m[:, :1], colors_precomp=m, scales=m, rotations=torch.zeros((1, 4), device=dev))


def test_prefiltered_trap_and_debug_mode(gpu_device, tmp_path, monkeypatch):
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = gpu_device
    monkeypatch.chdir(tmp_path)
    s = build_scene("culled_mix")
    st = _settings(s, dev, debug=True)._replace(prefiltered=True)
    m = torch.from_numpy(s.means3D).to(dev)
    with pytest.raises(RuntimeError, match="filtered although prefiltered"):
        GaussianRasterizer(st)(means3D=m, means2D=torch.zeros_like(m), shs=torch.from_numpy(s.shs).to(dev),
                               opacities=torch.from_numpy(s.opacities).to(dev).reshape(-1, 1),
                               scales=torch.from_numpy(s.scales).to(dev), rotations=torch.from_numpy(s.rotations).to(dev))
    assert (tmp_path / "snapshot_fw.dump").exists()   # reference behaviour with debug=True (__init__.py:83-90)


def test_determinism_of_forward(gpu_device):
    s = build_scene("capsule_circle")
    a, _ = util.run_product(s, gpu_device)
    b, _ = util.run_product(s, gpu_device)
    assert a["out_color"].tobytes() == b["out_color"].tobytes()
    np.testing.assert_array_equal(a["vals"], b["vals"])


def test_library_selftest_of_internal_primitives(gpu_device):
    """The matrix-core pixel contraction of the render backward (operand layout, block skipping) and the stable radix sort vs std::stable_sort."""
    from diff_gaussian_rasterization import _native as N
    N.selftest(gpu_device)


def test_hostile_inputs_fail_cleanly_or_render_finite(gpu_device):
    """Huge scales (every Gaussian touches every tile), zero / negative opacity, negative colours, NaN means: no hang,
    no out-of-bounds; either a clean error or a finite image for the finite part of the input."""
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = gpu_device
    s = build_scene("random_aniso")
    st = _settings(s, dev)
    m = torch.from_numpy(s.means3D).to(dev)
    sh, op = torch.from_numpy(s.shs).to(dev), torch.from_numpy(s.opacities).to(dev).reshape(-1, 1)
    sc, ro = torch.from_numpy(s.scales).to(dev), torch.from_numpy(s.rotations).to(dev)
    with torch.no_grad():
        # every Gaussian covers the whole (small) image: R = P * T, lists of P entries per tile
        img, radii = GaussianRasterizer(st)(means3D=m, means2D=torch.zeros_like(m), shs=sh, opacities=op * 0.05, scales=sc * 500, rotations=ro)
        assert img.isfinite().all()
        # opacity <= 0 never contributes: pure background
        img, _ = GaussianRasterizer(st)(means3D=m, means2D=torch.zeros_like(m), shs=sh, opacities=-op, scales=sc, rotations=ro)
        assert torch.equal(img, st.bg.reshape(3, 1, 1).expand_as(img))
        # NaN means are dropped by the near-plane test of the reference only if z compares false; they must not hang
        m2 = m.clone()
        m2[::7] = float("nan")
        img, radii = GaussianRasterizer(st)(means3D=m2, means2D=torch.zeros_like(m), shs=sh, opacities=op, scales=sc, rotations=ro)
        torch.cuda.synchronize()
        assert radii.shape == (s.P,)
    # a 1080p frame where all 3000 Gaussians are gigantic overflows nothing (3000 * 8160 pairs) but is refused above 2^31
    big = GaussianRasterizer(st._replace(image_height=8192, image_width=8192))
    P = 20000
    mm = torch.zeros((P, 3), device=dev)
    mm[:, 2] = 1.0
    with torch.no_grad(), pytest.raises(RuntimeError, match="does not fit the reference's int"):
        big(means3D=mm, means2D=torch.zeros_like(mm), colors_precomp=torch.ones((P, 3), device=dev),
            opacities=torch.ones((P, 1), device=dev), scales=torch.full((P, 3), 100.0, device=dev),
            rotations=torch.tensor([[1.0, 0, 0, 0]], device=dev).repeat(P, 1))


def test_concurrent_host_threads_get_the_sequential_result(gpu_device):
    """Frames in flight: several host threads, each on its own stream, render different scenes at once (with the
    per-stage profiler switched on and off from another thread meanwhile); every forward output must be byte-identical
    to the sequential run and the gradients within the usual tolerance of it."""
    import threading
    import torch
    from diff_gaussian_rasterization import _native as N
    names = ["random_aniso", "big_splats", "capsule_circle", "voxel_ties"]
    scenes = [util.build_scene(n) for n in names]
    dLs = [util.seeded_dL(s) for s in scenes]
    ref = [util.run_product(s, gpu_device, dL_dpix=d) for s, d in zip(scenes, dLs)]
    errs = []

    def worker(t):
        try:
            st = torch.cuda.Stream(device=gpu_device)
            with torch.cuda.stream(st):
                for it in range(12):
                    k = (t + it) % len(scenes)
                    p, g = util.run_product(scenes[k], gpu_device, dL_dpix=dLs[k])
                    rp, rg = ref[k]
                    assert p["R"] == rp["R"]
                    assert p["out_color"].tobytes() == rp["out_color"].tobytes(), names[k]
                    assert np.array_equal(p["vals"], rp["vals"]) and np.array_equal(p["n_contrib"], rp["n_contrib"])
                    for key in g:
                        if g[key].size:
                            scale = np.abs(rg[key]).max()
                            assert np.abs(g[key].astype(np.float64) - rg[key]).max() <= 2e-4 * scale + 1e-30, (names[k], key)
            st.synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    stop = threading.Event()

    def toggler():
        while not stop.is_set():
            N.set_profiling(True)
            N.get_profile()
            N.set_profiling(False)

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    tg = threading.Thread(target=toggler)
    tg.start()
    for x in ths:
        x.start()
    for x in ths:
        x.join()
    stop.set()
    tg.join()
    N.set_profiling(False)
    assert not errs, errs[0]


def test_float_aligned_gradient_outputs_take_the_scalar_store_path(gpu_device):
    """the per-Gaussian backward writes its 3- and 6-float rows (and dL_dsh) as 16-byte chunks from each
    array's base, which needs 16-byte aligned outputs (include/gsr.h); a C-ABI caller may hand over float-aligned
    sub-buffers: those are written through per-row stores -- same numbers -- and only dL_drot (one float4 per Gaussian) is
    refused when misaligned."""
    from diff_gaussian_rasterization import _native as N
    dev = gpu_device
    s = util.build_scene("sh_deg2")
    dL = util.seeded_dL(s)

    def t(a):
        return torch.empty(0) if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    args = (t(s.bg), t(s.means3D), t(s.colors_precomp), t(s.opacities), t(s.scales), t(s.rotations), s.scale_modifier,
            t(s.cov3D_precomp), t(s.viewmatrix.reshape(4, 4)), t(s.projmatrix.reshape(4, 4)), s.tanfovx, s.tanfovy, s.H, s.W,
            t(s.shs), s.sh_degree, t(s.campos), s.prefiltered, False)
    R, color, radii, geom, binning, img = N.rasterize_gaussians(*args, need_backward=True)
    bargs = (args[0], args[1], radii, args[2], args[4], args[5], s.scale_modifier, args[7], args[8], args[9], s.tanfovx, s.tanfovy,
             t(dL), args[14], s.sh_degree, args[16], geom, R, binning, img, False)
    want = N.rasterize_gaussians_backward(*bargs)

    def shifted(shape, dtype=None, device=None):      # 4 bytes past a 16-byte boundary, except the [P,4] rotation gradient
        n = int(np.prod(shape))
        if len(shape) == 2 and shape[1] == 4:
            return torch.empty(shape, dtype=dtype, device=device)
        return torch.empty((n + 1,), dtype=dtype, device=device)[1:].view(shape)
    got = N.rasterize_gaussians_backward_batch(
        bargs[0], bargs[1], radii.reshape(1, -1), bargs[3], bargs[4], bargs[5], bargs[6], bargs[7], bargs[8].reshape(1, 4, 4),
        bargs[9].reshape(1, 4, 4), bargs[10], bargs[11], bargs[12].reshape(1, 3, s.H, s.W), bargs[13], bargs[14],
        bargs[15].reshape(1, 3), geom, binning, img, False, _alloc=shifted)
    assert got[0].data_ptr() % 16 == 4
    for a, b in zip(got, want):
        if a.numel():
            scale = float(b.abs().max()) + 1e-30
            assert float((a - b).abs().max()) <= 2e-5 * scale      # atomics commit in another order between two backwards

    def all_shifted(shape, dtype=None, device=None):
        return torch.empty((int(np.prod(shape)) + 1,), dtype=dtype, device=device)[1:].view(shape)
    with pytest.raises(RuntimeError, match="dL_drot must be 16-byte aligned"):
        N.rasterize_gaussians_backward_batch(
            bargs[0], bargs[1], radii.reshape(1, -1), bargs[3], bargs[4], bargs[5], bargs[6], bargs[7], bargs[8].reshape(1, 4, 4),
            bargs[9].reshape(1, 4, 4), bargs[10], bargs[11], bargs[12].reshape(1, 3, s.H, s.W), bargs[13], bargs[14],
            bargs[15].reshape(1, 3), geom, binning, img, False, _alloc=all_shifted)


def test_short_host_path_equals_the_general_path_and_survives_a_grown_frame(gpu_device):
    """Steady-state per-view calls take _native.forward_view / backward_view (one arena allocation, inline checks: the host time
    between the caller's last copy and the first kernel).  Same images, radii and gradients as the general path (debug=True settings
    never take the short one); a frame that suddenly needs more pairs than the remembered capacity repeats its binning half on a
    larger arena (GSR_RETRY inside the short path) and still renders the right image."""
    from diff_gaussian_rasterization import GaussianRasterizer, _native as N
    import diff_gaussian_rasterization as d
    dev = gpu_device
    s = build_scene("capsule_circle")
    dL = torch.from_numpy(seeded_dL(s)).to(dev)
    taken = []
    real = N.forward_view

    def spy(*a, **k):
        r = real(*a, **k)
        taken.append(r is not None)
        return r

    def run(scales_factor, debug):
        L = dict(means3D=_leaf(s.means3D, dev), shs=_leaf(s.shs, dev), opacities=_leaf(s.opacities.reshape(-1, 1), dev),
                 scales=_leaf(s.scales * scales_factor, dev), rotations=_leaf(s.rotations, dev))
        L["means2D"] = torch.zeros_like(L["means3D"], requires_grad=True)
        img, radii = GaussianRasterizer(_settings(s, dev, debug=debug))(**L)
        (img * dL).sum().backward()
        return img.detach(), radii, {k: v.grad for k, v in L.items()}

    N.forward_view = spy
    try:
        N.reset_capacity_hints()
        run(1.0, False)                        # first frame of the configuration: counts synchronously (general path)
        a = run(1.0, False)                    # steady state: the short path
        assert taken[-2:] == [False, True]
        b = run(1.0, True)                     # debug settings: the general path
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        for k in a[2]:
            ga, gb = a[2][k], b[2][k]
            assert (ga - gb).abs().max() <= 2e-4 * gb.abs().max() + 1e-30, k          # (atomic order)
        # three times the scales: ~9x the pairs, far beyond capacity hint x 1.25 -> the short path retries
        hint = N._CAP_HINT[N._cap_key(dev, s.P, s.W, s.H)]
        c = run(3.0, False)
        assert taken[-1] is True and N._CAP_HINT[N._cap_key(dev, s.P, s.W, s.H)] > 2 * hint
        e = run(3.0, True)
        assert torch.equal(c[0], e[0]) and torch.equal(c[1], e[1])
        for k in c[2]:
            assert (c[2][k] - e[2][k]).abs().max() <= 2e-4 * e[2][k].abs().max() + 1e-30, k
    finally:
        N.forward_view = real
    assert d._C.forward_view is real
