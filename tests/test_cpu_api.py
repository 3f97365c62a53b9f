"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/gsr.h declares, the Python
mirror keeps the reference's surface, host-side validation behaves like the reference's, and there is no CPU
compute path (no compute calls are made here: the container has no GPU)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from diff_gaussian_rasterization import _native
    header = open(os.path.join(ROOT, "include", "gsr.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(_native.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), "libgsr_hip.so does not export %s" % sym
    assert sorted(_native.SYMBOLS) == declared
    assert b"gfx950" in _native.lib.gsr_version()


def test_arena_size_queries():
    from diff_gaussian_rasterization import _native
    lib = _native.lib
    prev = 0
    for P in (0, 1, 255, 256, 4097, 800_000, 2_000_000):
        b = lib.gsr_geom_bytes(P)
        assert b >= prev and b % 256 == 0
        prev = b
    # SoA arena: 64-B splat line + 64-B gradient record + depth keys/ids/tile count/clamp mask per Gaussian -> ~149 B/Gaussian
    # + per-workgroup histograms and prefix-sum status words
    assert 140 * 800_000 < lib.gsr_geom_bytes(800_000) < 160 * 800_000
    # binning: two key + two u32 id buffers (16 B/pair) + the backward pass's chunk-boundary state (4 KB per 512
    # pairs = 8 B/pair: carved for the shortest chunk length, the one single-view submissions use); the reference needs
    # 24 B/pair + CUB temp -- for 1.5x the pairs (footprint clipping)
    assert 24 * 11_500_000 <= lib.gsr_binning_bytes(11_500_000) < 25 * 11_500_000
    assert lib.gsr_binning_bytes(0) > 0
    assert lib.gsr_image_bytes(1920, 1080) >= 20 * 1920 * 1080 + 8 * 8160
    # sizes that do not fit int32 pair counts are still answered
    assert lib.gsr_binning_bytes(87_000_000) > 16 * 87_000_000


def test_python_surface_matches_reference():
    import diff_gaussian_rasterization as d
    assert d.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")
    import inspect
    sig = inspect.signature(d.GaussianRasterizer.forward)
    assert list(sig.parameters) == ["self", "means3D", "means2D", "opacities", "shs", "colors_precomp", "scales",
                                    "rotations", "cov3D_precomp"]
    assert list(inspect.signature(d.rasterize_gaussians).parameters) == [
        "means3D", "means2D", "sh", "colors_precomp", "opacities", "scales", "rotations", "cov3Ds_precomp", "raster_settings"]
    assert issubclass(d.GaussianRasterizer, torch.nn.Module) and hasattr(d.GaussianRasterizer, "markVisible")
    assert issubclass(d._RasterizeGaussians, torch.autograd.Function)
    t = d.cpu_deep_copy_tuple((torch.ones(2), 3, "x"))
    assert t[1] == 3 and t[2] == "x" and torch.equal(t[0], torch.ones(2))


def _settings():
    import diff_gaussian_rasterization as d
    return d.GaussianRasterizationSettings(
        image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3), scale_modifier=1.0,
        viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0, campos=torch.zeros(3), prefiltered=False,
        debug=False)


def test_argument_validation_is_the_reference_s():
    import diff_gaussian_rasterization as d
    r = d.GaussianRasterizer(_settings())
    m = torch.zeros(2, 3)
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        r(means3D=m, means2D=m, opacities=m[:, :1])
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        r(means3D=m, means2D=m, opacities=m[:, :1], shs=m[:, None], colors_precomp=m)
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=m[:, :1], colors_precomp=m, scales=m)
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=m[:, :1], colors_precomp=m, scales=m, rotations=torch.zeros(2, 4),
          cov3D_precomp=torch.zeros(2, 6))


def test_no_cpu_fallback():
    """CPU tensors must be refused loudly; the product never routes through the oracle or any CPU code."""
    import diff_gaussian_rasterization as d
    r = d.GaussianRasterizer(_settings())
    m = torch.zeros(2, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(means3D=m, means2D=m, opacities=m[:, :1], colors_precomp=m, scales=m, rotations=torch.zeros(2, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        r.markVisible(m)
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):
        r(means3D=torch.zeros(6), means2D=m, opacities=m[:, :1], colors_precomp=m, scales=m, rotations=torch.zeros(2, 4))
    # nothing under the product package imports the oracle
    pkg = os.path.join(ROOT, "gaussian-pcloud-render_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "from oracle" not in src and "import oracle" not in src and "gsr_oracle" not in src and "_ref/" not in src, f


def test_c_abi_rejects_bad_arguments_without_a_gpu():
    """Argument checks happen before any HIP call, so they can be exercised on the CPU-only container."""
    from diff_gaussian_rasterization import _native as N
    p = N.GsrParams()
    p.P, p.W, p.H = 10, 0, 8
    R = ctypes.c_int64(-1)
    rc = N.lib.gsr_forward_stage1(ctypes.byref(p), None, 0, None, 0, None, ctypes.byref(R), None)
    assert rc == -1 and b"bad sizes" in N.lib.gsr_last_error()
    p.W = 8
    rc = N.lib.gsr_forward_stage1(ctypes.byref(p), None, 0, None, 0, None, ctypes.byref(R), None)
    assert rc == -1 and b"required input pointer is NULL" in N.lib.gsr_last_error()
    p.P = 0   # empty cloud: nothing to do, success, zero pairs
    rc = N.lib.gsr_forward_stage1(ctypes.byref(p), None, 0, None, 0, None, ctypes.byref(R), None)
    assert rc == 0 and R.value == 0
    assert N.lib.gsr_mark_visible(-1, None, None, None, None, None) == -1
    assert N.lib.gsr_mark_visible(0, None, None, None, None, None) == 0


def test_forward_batch_rejects_bad_arguments_without_a_gpu():
    from diff_gaussian_rasterization import _native as N
    p = N.GsrParams()
    p.P, p.W, p.H = 10, 8, 8
    cnt = (ctypes.c_int64 * 4)()
    rc = N.lib.gsr_forward_batch(ctypes.byref(p), 0, None, 0, None, 0, None, 0, None, None, cnt, 0, None)
    assert rc == -1 and b"view count" in N.lib.gsr_last_error()
    rc = N.lib.gsr_forward_batch(ctypes.byref(p), 4, None, 0, None, 0, None, 0, None, None, cnt, 0, None)
    assert rc == -1 and b"required input pointer is NULL" in N.lib.gsr_last_error()
    p.P = 0
    cnt[2] = 7
    rc = N.lib.gsr_forward_batch(ctypes.byref(p), 4, None, 0, None, 0, None, 0, None, None, cnt, 0, None)
    assert rc == 0 and list(cnt) == [0, 0, 0, 0]
    # an image with more than 2^28 tiles cannot be described by the backward's work items
    p.P, p.W, p.H = 10, 16 * 20000, 16 * 20000
    rc = N.lib.gsr_forward_batch(ctypes.byref(p), 1, None, 0, None, 0, None, 0, None, None, cnt, 0, None)
    assert rc == -1


def test_rasterize_views_validates_like_the_reference():
    import diff_gaussian_rasterization as d
    m = torch.zeros(2, 3)
    st = _settings()
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        d.rasterize_views(m, m, m[:, :1], [st, st])
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        d.rasterize_views(m, m, m[:, :1], [st], colors_precomp=m, scales=m)
    with pytest.raises(RuntimeError, match="no CPU path"):
        d.rasterize_views(m, m, m[:, :1], [st, st], colors_precomp=m, scales=m, rotations=torch.zeros(2, 4))
    st2 = st._replace(image_width=16)
    with pytest.raises(Exception, match="must share image size"):
        d.rasterize_views(m, m, m[:, :1], [st, st2], colors_precomp=m, scales=m, rotations=torch.zeros(2, 4))


def test_a_missing_reference_build_fails_the_gpu_comparisons(monkeypatch):
    """The -m gpu tests that compare with oracle/_ref must turn red, not yellow, when the .so did not travel to the GPU box."""
    import _pytest.outcomes
    import util
    from oracle import oracle as orc
    monkeypatch.setitem(orc.REF_SO, "strict", os.path.join(ROOT, "oracle", "_ref", "no_such_library.so"))
    monkeypatch.delenv("GSR_ALLOW_NO_REF", raising=False)
    with pytest.raises(_pytest.outcomes.Failed):
        util.reference_build("strict")
    monkeypatch.setenv("GSR_ALLOW_NO_REF", "1")
    with pytest.raises(_pytest.outcomes.Skipped):
        util.reference_build("strict")
    monkeypatch.setenv("GSR_REQUIRE_REF", "1")
    with pytest.raises(_pytest.outcomes.Failed):
        util.reference_build("strict")
