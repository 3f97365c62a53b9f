"""GPU tests (-m gpu) of the HOST side of the call path: nothing between the caller and the kernels may wait for the device
or copy to the host, except the one 32-byte counter read-back per forward call that the C ABI documents (include/gsr.h).

  * GaussianRasterizer.forward / backward and rasterize_views under torch's sync-debug mode "error" (any torch-level
    synchronising op -- .cpu(), .item(), torch.equal on device tensors -- raises), with the library's own read-back counter
    advancing by exactly one per forward call and not at all per backward;
  * the packed per-view blocks of rasterize_views are built once per settings list, not per call;
  * a repeated backward over one forward (retain_graph=True; two gsr_backward_batch calls on the same arenas) returns the
    same gradients each time (the reference zero-fills its accumulators per call, rasterize_points.cu:151-159).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _setup(dev, n_views=3, P=12000, W=208, H=176):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    from pcrender import camera, synth
    cloud = synth.make_cloud("synth-THuman-256", seed=0, P=P)
    g = synth.make_gaussians(cloud, profile="training", seed=1)
    views = camera.circle_views(12, fov_deg=45.0, width_px=W, height_px=H)[:n_views]
    bg = torch.ones(3, device=dev)
    settings = [GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=v["tanfovx"], tanfovy=v["tanfovy"], bg=bg, scale_modifier=1.0,
        viewmatrix=v["viewmatrix"].to(dev), projmatrix=v["projmatrix"].to(dev), sh_degree=g["sh_degree"],
        campos=v["campos"].to(dev), prefiltered=False, debug=False) for v in views]

    def leaves():
        m3 = _t(g["means3D"], dev).requires_grad_(True)
        return dict(means3D=m3, means2D=torch.zeros_like(m3, requires_grad=True), shs=_t(g["shs"], dev).requires_grad_(True),
                    opacities=_t(g["opacities"], dev).requires_grad_(True), scales=_t(g["scales"], dev).requires_grad_(True),
                    rotations=_t(g["rotations"], dev).requires_grad_(True))
    G = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, (3, H, W)).astype(np.float32)).to(dev)
    return settings, leaves, G


class _no_sync:
    """torch raises on every synchronising torch call inside the block"""
    def __enter__(self):
        self.old = torch.cuda.get_sync_debug_mode()
        torch.cuda.set_sync_debug_mode("error")

    def __exit__(self, *a):
        torch.cuda.set_sync_debug_mode(self.old)


def test_per_view_call_makes_one_readback_and_no_torch_sync(gpu_device):
    from diff_gaussian_rasterization import GaussianRasterizer, _native as N
    dev = gpu_device
    settings, leaves, G = _setup(dev)
    L = leaves()
    r = GaussianRasterizer(settings[1])
    for _ in range(2):      # first call of a configuration counts pairs synchronously (stage 1 / stage 2); then steady state
        img, _ = r(**L)
        (img * G).sum().backward()
    torch.cuda.synchronize()
    for rep in range(3):
        c0 = N.lib.gsr_d2h_count()
        with _no_sync():
            img, radii = r(**L)
            c1 = N.lib.gsr_d2h_count()
            (img * G).sum().backward()
            c2 = N.lib.gsr_d2h_count()
        assert c1 - c0 == 1, "forward: %d device->host read-backs (exactly one expected)" % (c1 - c0)
        assert c2 - c1 == 0, "backward must not read anything back"
    torch.cuda.synchronize()
    assert torch.isfinite(L["means3D"].grad).all()


def test_rasterize_views_makes_one_readback_and_no_torch_sync(gpu_device):
    from diff_gaussian_rasterization import _native as N, rasterize_views
    import diff_gaussian_rasterization as d
    dev = gpu_device
    settings, leaves, G = _setup(dev, n_views=4)
    L = leaves()

    def call():
        imgs, radii = rasterize_views(L["means3D"], L["means2D"], L["opacities"], settings, shs=L["shs"], scales=L["scales"],
                                      rotations=L["rotations"])
        (imgs * G).sum().backward()
        return imgs

    for _ in range(2):
        call()
    torch.cuda.synchronize()
    blocks = len(d._VIEW_BLOCKS)
    for rep in range(3):
        c0 = N.lib.gsr_d2h_count()
        with _no_sync():
            call()
        assert N.lib.gsr_d2h_count() - c0 == 1
    # a NEW list object of the same settings reuses the packed blocks too
    with _no_sync():
        imgs2, _ = rasterize_views(L["means3D"], L["means2D"], L["opacities"], list(settings), shs=L["shs"], scales=L["scales"],
                                   rotations=L["rotations"])
    assert len(d._VIEW_BLOCKS) == blocks
    torch.cuda.synchronize()
    # in-place change of a view matrix is seen (version counter in the key), not served from the cache
    ref = imgs2.detach().clone()
    settings[2].viewmatrix.mul_(1.0)
    imgs3, _ = rasterize_views(L["means3D"], L["means2D"], L["opacities"], settings, shs=L["shs"], scales=L["scales"],
                               rotations=L["rotations"])
    assert len(d._VIEW_BLOCKS) == blocks + 1 and torch.equal(imgs3, ref)
    # backgrounds that are different tensors with equal values are accepted, unequal ones rejected
    s_alt = [s._replace(bg=torch.ones(3, device=dev)) if i == 1 else s for i, s in enumerate(settings)]
    rasterize_views(L["means3D"], L["means2D"], L["opacities"], s_alt, shs=L["shs"], scales=L["scales"], rotations=L["rotations"])
    s_bad = [s._replace(bg=torch.zeros(3, device=dev)) if i == 1 else s for i, s in enumerate(settings)]
    with pytest.raises(Exception, match="must share image size"):
        rasterize_views(L["means3D"], L["means2D"], L["opacities"], s_bad, shs=L["shs"], scales=L["scales"], rotations=L["rotations"])


def test_view_block_cache_keys_on_strides_and_survives_inference_tensors(gpu_device):
    """(1) cameras built under torch.inference_mode() have no version counter -- rasterize_views must not
    crash, it packs their blocks per call; (2) M and M.T share address and version but not strides: they are different
    cameras and must not collide on one cache entry; (3) set_view_cache(False) packs on every call (the remedy for writes
    that bypass the version counter, `m.data.copy_`)."""
    from diff_gaussian_rasterization import rasterize_views
    import diff_gaussian_rasterization as d
    dev = gpu_device
    settings, leaves, G = _setup(dev, n_views=2)
    L = leaves()

    def render(sl):
        with torch.no_grad():
            return rasterize_views(L["means3D"], L["means2D"], L["opacities"], sl, shs=L["shs"], scales=L["scales"],
                                   rotations=L["rotations"])[0].clone()
    ref = render(settings)
    # (1) inference tensors
    with torch.inference_mode():
        s_inf = [s._replace(viewmatrix=s.viewmatrix.clone(), projmatrix=s.projmatrix.clone(), campos=s.campos.clone())
                 for s in settings]
    n0 = len(d._VIEW_BLOCKS)
    assert torch.equal(render(s_inf), ref)
    assert len(d._VIEW_BLOCKS) == n0, "a settings list with inference tensors must bypass the cache"
    # (2) a transposed alias of the same storage is another camera
    m = settings[0].viewmatrix.reshape(4, 4)
    s_t = [settings[0]._replace(viewmatrix=m.t()), settings[1]]
    want = render([settings[0]._replace(viewmatrix=m.t().contiguous()), settings[1]])
    render(settings)                       # make sure the untransposed entry is the cached one
    got = render(s_t)
    assert torch.equal(got, want) and not torch.equal(got, ref)
    # (3) cache off: a write that bypasses the version counter is seen
    d.set_view_cache(False)
    try:
        s_w = [s._replace(viewmatrix=s.viewmatrix.clone()) for s in settings]
        a = render(s_w)
        s_w[0].viewmatrix.data.copy_(m.t().contiguous().reshape(s_w[0].viewmatrix.shape))
        b = render(s_w)
        assert torch.equal(a, ref) and torch.equal(b, want)
        assert len(d._VIEW_BLOCKS) == 0
    finally:
        d.set_view_cache(True)


def test_repeated_backward_returns_the_same_gradients(gpu_device):
    from diff_gaussian_rasterization import GaussianRasterizer, rasterize_views
    dev = gpu_device
    settings, leaves, G = _setup(dev, n_views=3)
    # per-view call, retain_graph
    L = leaves()
    img, _ = GaussianRasterizer(settings[1])(**L)
    loss = (img * G).sum()
    names = list(L)
    g1 = torch.autograd.grad(loss, [L[k] for k in names], retain_graph=True)
    g2 = torch.autograd.grad(loss, [L[k] for k in names], retain_graph=True)
    g3 = torch.autograd.grad(2.0 * loss, [L[k] for k in names])
    for k, a, b, c in zip(names, g1, g2, g3):
        scale = float(a.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 1e-5 * scale, "second backward differs from the first (%s)" % k
        assert float((c - 2 * a).abs().max()) <= 1e-5 * 2 * scale, "third backward (2 x loss) is not twice the first (%s)" % k
    # view batch
    L = leaves()
    imgs, _ = rasterize_views(L["means3D"], L["means2D"], L["opacities"], settings, shs=L["shs"], scales=L["scales"],
                              rotations=L["rotations"])
    loss = (imgs * G).sum()
    g1 = torch.autograd.grad(loss, [L[k] for k in names], retain_graph=True)
    g2 = torch.autograd.grad(loss, [L[k] for k in names])
    for k, a, b in zip(names, g1, g2):
        assert float((a - b).abs().max()) <= 1e-5 * (float(a.abs().max()) + 1e-30), "batch: second backward differs (%s)" % k


def test_inference_geometry_arena_is_smaller_and_enough(gpu_device):
    from diff_gaussian_rasterization import _native as N
    dev = gpu_device
    settings, leaves, G = _setup(dev, n_views=3)
    L = leaves()
    P = L["means3D"].shape[0]
    assert N.lib.gsr_geom_bytes(P) - N.lib.gsr_geom_bytes_inference(P) >= 64 * P
    e = torch.empty(0)
    vm = torch.stack([s.viewmatrix for s in settings]); pm = torch.stack([s.projmatrix for s in settings])
    cp = torch.stack([s.campos.reshape(3) for s in settings])
    s0 = settings[0]
    args = (s0.bg, L["means3D"].detach(), e, L["opacities"].detach(), L["scales"].detach(), L["rotations"].detach(), 1.0, e, vm, pm,
            s0.tanfovx, s0.tanfovy, s0.image_height, s0.image_width, L["shs"].detach(), s0.sh_degree, cp, False, False)
    c_inf, col_inf, rad_inf, geom_inf, *_ = N.rasterize_gaussians_batch(*args, need_backward=False)
    c_bwd, col_bwd, rad_bwd, geom_bwd, *_ = N.rasterize_gaussians_batch(*args, need_backward=True)
    assert geom_inf.numel() == 3 * N.lib.gsr_geom_bytes_inference(P) and geom_bwd.numel() == 3 * N.lib.gsr_geom_bytes(P)
    assert c_inf == c_bwd and torch.equal(col_inf, col_bwd) and torch.equal(rad_inf, rad_bwd)


def test_consecutive_per_view_calls_overlap_only_when_the_inputs_are_provably_unchanged(gpu_device):
    """The library starts forward(j+1) beside backward(j) when every input is a tensor object it has seen before with an
    unchanged version counter (_native._OnSideStream).  (1) Steady state of a per-view training loop: calls overlap and give
    the gradients of the in-order run.  (2) An in-place update between two calls (an optimizer step) bumps the version
    counter: the next call waits for the caller's stream and renders the UPDATED cloud.  (3) Fresh tensors every call
    (matrices rebuilt on the device, like the reference's caller): never overlapped, same images.  (4) Switched off: every
    kernel on the caller's stream again."""
    from diff_gaussian_rasterization import GaussianRasterizer, _native as N
    dev = gpu_device
    settings, leaves, G = _setup(dev, n_views=4)

    def loop(L, sl, n):
        imgs = []
        for k in range(n):
            img, _ = GaussianRasterizer(sl[k % len(sl)])(**L)
            (img * G).sum().backward()
            imgs.append(img.detach().clone())
        torch.cuda.synchronize()
        return imgs

    was_on = N._OVERLAP_ON          # (off by default since round 5: opt-in, GSR_OVERLAP=1)
    N.set_overlap(False)
    L0 = leaves()
    want = loop(L0, settings, 8)
    g_want = {k: v.grad.clone() for k, v in L0.items()}
    N.set_overlap(True)
    N.OVERLAP_STATS.update(calls=0, overlapped=0)
    L1 = leaves()
    got = loop(L1, settings, 8)
    assert N.OVERLAP_STATS["calls"] == 8 and N.OVERLAP_STATS["overlapped"] >= 3, N.OVERLAP_STATS   # second turn of the 4 views
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    for k in L1:
        scale = float(g_want[k].abs().max()) + 1e-30
        assert float((L1[k].grad - g_want[k]).abs().max()) <= 2e-5 * scale, k
    # (2) an in-place step between calls is seen
    N.OVERLAP_STATS.update(calls=0, overlapped=0)
    with torch.no_grad():
        L1["means3D"].add_(0.01)
    img_moved, _ = GaussianRasterizer(settings[0])(**L1)
    assert N.OVERLAP_STATS["overlapped"] == 0
    L2 = leaves()
    with torch.no_grad():
        L2["means3D"].add_(0.01)
    N.set_overlap(False)
    img_ref, _ = GaussianRasterizer(settings[0])(**L2)
    assert torch.equal(img_moved, img_ref) and not torch.equal(img_moved, want[0])
    # (3) fresh matrices every call
    N.set_overlap(True)
    N.OVERLAP_STATS.update(calls=0, overlapped=0)
    with torch.no_grad():
        for k in range(6):
            s = settings[k % 4]
            s2 = s._replace(viewmatrix=s.viewmatrix * 1.0, projmatrix=s.projmatrix * 1.0, campos=s.campos * 1.0)
            img, _ = GaussianRasterizer(s2)(**L0)
            assert torch.equal(img, want[k % 4])
    assert N.OVERLAP_STATS["overlapped"] == 0 and N.OVERLAP_STATS["calls"] == 6
    # (4) a storage swapped under the same tensor object (`param.data = other`: no version bump) is seen as a new input
    N.OVERLAP_STATS.update(calls=0, overlapped=0)
    L3 = leaves()
    with torch.no_grad():
        GaussianRasterizer(settings[0])(**L3)
        GaussianRasterizer(settings[1])(**L3)
        assert N.OVERLAP_STATS["overlapped"] == 1
        moved = L3["means3D"].detach() + 0.01          # produced on the caller's stream right before the call
        ver = L3["means3D"]._version
        L3["means3D"].data = moved
        assert L3["means3D"]._version == ver
        img_sw, _ = GaussianRasterizer(settings[0])(**L3)
    assert N.OVERLAP_STATS["overlapped"] == 1 and torch.equal(img_sw, img_ref)
    N.set_overlap(was_on)


def test_overlap_of_consecutive_calls_survives_a_random_training_script(gpu_device):
    """A scripted random mix of what a caller can do between two per-view calls -- optimizer-like in-place steps, a parameter
    REPLACED by a new tensor, forward-only calls, calls under another current stream, retain_graph backwards, freed results --
    gives bit-identical images and (to atomic-order noise) identical accumulated gradients with the library's overlap of
    consecutive calls on and off."""
    from diff_gaussian_rasterization import GaussianRasterizer, _native as N
    dev = gpu_device
    settings, leaves, G = _setup(dev, n_views=5, P=9000, W=176, H=144)
    other = torch.cuda.Stream(device=dev)

    def script(seed):
        rng = np.random.default_rng(seed)
        L = leaves()
        sums, keep = [], []
        for step in range(70):
            v = int(rng.integers(0, len(settings)))
            what = rng.random()
            if what < 0.15:                      # an optimizer step on a random parameter (in place, under no_grad)
                k = ["means3D", "opacities", "scales", "shs"][int(rng.integers(0, 4))]
                with torch.no_grad():
                    L[k].add_(1e-3 * torch.from_numpy(rng.standard_normal(tuple(L[k].shape)).astype(np.float32)).to(dev))
            elif what < 0.22:                    # a parameter replaced by a NEW tensor (densification does this)
                L["rotations"] = (L["rotations"].detach() * 1.0).requires_grad_(True)
            elif what < 0.29:                    # the STORAGE replaced under the same tensor object: no version bump (weight
                L["opacities"].data = (L["opacities"].detach() * 0.999).clamp_(0.01, 1.0)   # clamping, checkpoint loading)
            ctx = torch.cuda.stream(other) if rng.random() < 0.2 else _Null()
            if isinstance(ctx, _Null):
                pass
            else:
                other.wait_stream(torch.cuda.current_stream(dev))
            with ctx:
                if rng.random() < 0.25:
                    with torch.no_grad():
                        img, _ = GaussianRasterizer(settings[v])(**L)
                else:
                    img, _ = GaussianRasterizer(settings[v])(**L)
                    loss = (img * G).sum()
                    if rng.random() < 0.2:
                        loss.backward(retain_graph=True)
                    loss.backward()
                sums.append(img.detach().clone())
                if rng.random() < 0.3:
                    keep.append(img)             # some results stay alive, most are freed at once
            if not isinstance(ctx, _Null):
                torch.cuda.current_stream(dev).wait_stream(other)
        torch.cuda.synchronize()
        return sums, {k: (None if t.grad is None else t.grad.clone()) for k, t in L.items()}

    class _Null:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    was_on = N._OVERLAP_ON
    N.set_overlap(False)
    want, gw = script(11)
    N.set_overlap(True)
    N.OVERLAP_STATS.update(calls=0, overlapped=0)
    got, gg = script(11)
    N.set_overlap(was_on)
    assert N.OVERLAP_STATS["overlapped"] > 10, N.OVERLAP_STATS
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), "step %d" % i
    for k in gw:
        if gw[k] is None:
            assert gg[k] is None
            continue
        scale = float(gw[k].abs().max()) + 1e-30
        assert float((gg[k] - gw[k]).abs().max()) <= 1e-4 * scale, k
