"""ctypes binding of libgsr_hip.so (include/gsr.h).

Plays the role of the reference's pybind module ``diff_gaussian_rasterization._C`` (ext.cpp:15-19):
three entry points taking torch tensors.  Tensors are only used for device memory and the
stream; what crosses the boundary are raw pointers and sizes.

There is NO fallback: if the shared library is missing or a tensor is not on a HIP device the
call raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgsr_hip.so")

_fp = C.c_void_p


class GsrParams(C.Structure):
    _fields_ = [
        ("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("W", C.c_int), ("H", C.c_int),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("prefiltered", C.c_int), ("debug", C.c_int), ("need_backward", C.c_int),
        ("bg", _fp), ("means3D", _fp), ("shs", _fp), ("colors_precomp", _fp), ("opacities", _fp),
        ("scales", _fp), ("rotations", _fp), ("cov3D_precomp", _fp),
        ("viewmatrix", _fp), ("projmatrix", _fp), ("campos", _fp),
    ]


# every symbol include/gsr.h declares (tests check the library exports all of them)
SYMBOLS = ("gsr_geom_bytes", "gsr_image_bytes", "gsr_binning_bytes", "gsr_forward_stage1", "gsr_forward_stage2",
           "gsr_backward", "gsr_mark_visible", "gsr_query", "gsr_set_profiling", "gsr_get_profile", "gsr_last_error",
           "gsr_version", "gsr_selftest", "gsr_forward_recolor")

Q = dict(DEPTHS=1, MEANS2D=2, CONIC_OPACITY=3, RGB=4, TILES_TOUCHED=5, POINT_LIST=6, POINT_LIST_KEYS=7, RANGES=8,
         FINAL_T=9, N_CONTRIB=10, CLAMPED=11, TILE_NEED=12)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "diff_gaussian_rasterization: %s not found - build it with "
            "`python gaussian-pcloud-render_amd/build.py` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.gsr_geom_bytes.restype = C.c_size_t
    lib.gsr_geom_bytes.argtypes = [C.c_int]
    lib.gsr_image_bytes.restype = C.c_size_t
    lib.gsr_image_bytes.argtypes = [C.c_int, C.c_int]
    lib.gsr_binning_bytes.restype = C.c_size_t
    lib.gsr_binning_bytes.argtypes = [C.c_int64]
    lib.gsr_forward_stage1.restype = C.c_int
    lib.gsr_forward_stage1.argtypes = [C.POINTER(GsrParams), _fp, C.c_size_t, _fp, C.c_size_t, _fp,
                                       C.POINTER(C.c_int64), _fp]
    lib.gsr_forward_stage2.restype = C.c_int
    lib.gsr_forward_stage2.argtypes = [C.POINTER(GsrParams), _fp, C.c_size_t, _fp, C.c_size_t, _fp, C.c_size_t,
                                       C.c_int64, _fp, _fp]
    lib.gsr_forward_recolor.restype = C.c_int
    lib.gsr_forward_recolor.argtypes = [C.POINTER(GsrParams), _fp, C.c_size_t, _fp, C.c_size_t, _fp, C.c_size_t, C.c_int64, _fp, _fp]
    lib.gsr_backward.restype = C.c_int
    lib.gsr_backward.argtypes = [C.POINTER(GsrParams), _fp, C.c_int64, _fp, C.c_size_t, _fp, C.c_size_t, _fp,
                                 C.c_size_t] + [_fp] * 10 + [_fp]
    lib.gsr_mark_visible.restype = C.c_int
    lib.gsr_mark_visible.argtypes = [C.c_int, _fp, _fp, _fp, _fp, _fp]
    lib.gsr_query.restype = C.c_int
    lib.gsr_query.argtypes = [C.POINTER(GsrParams), C.c_int, _fp, _fp, _fp, C.c_int64, _fp, C.c_size_t, _fp]
    lib.gsr_set_profiling.restype = None
    lib.gsr_set_profiling.argtypes = [C.c_int]
    lib.gsr_get_profile.restype = C.c_int
    lib.gsr_get_profile.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
    lib.gsr_selftest.restype = C.c_int
    lib.gsr_selftest.argtypes = [_fp]
    lib.gsr_last_error.restype = C.c_char_p
    lib.gsr_version.restype = C.c_char_p
    return lib


lib = _load()


def _ptr(t):
    """NULL for absent optionals: the reference passes empty tensors whose data_ptr is nullptr
    (rasterize_points.cu:94-111 -> rasterizer_impl.cu:321,389,411)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _check(rc):
    if rc != 0:
        raise RuntimeError(lib.gsr_last_error().decode("utf-8", "replace"))


def _f32c(t, device, name):
    if t.numel() == 0:
        return t
    if t.dtype != torch.float32:
        raise RuntimeError("expected scalar type Float but found %s for %s" % (t.dtype, name))
    return t.to(device).contiguous()


def _require_hip(device):
    if device.type != "cuda":
        raise RuntimeError(
            "diff_gaussian_rasterization (MI355X build) needs tensors on a HIP device (torch device 'cuda'); got %s. "
            "There is no CPU path." % device)


def _params(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
            tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, debug, need_backward):
    device = means3D.device
    keep = dict(
        bg=_f32c(bg, device, "bg"), means3D=_f32c(means3D, device, "means3D"), shs=_f32c(sh, device, "sh"),
        colors_precomp=_f32c(colors, device, "colors_precomp"), opacities=_f32c(opacity, device, "opacities"),
        scales=_f32c(scales, device, "scales"), rotations=_f32c(rotations, device, "rotations"),
        cov3D_precomp=_f32c(cov3D_precomp, device, "cov3D_precomp"),
        viewmatrix=_f32c(viewmatrix, device, "viewmatrix"), projmatrix=_f32c(projmatrix, device, "projmatrix"),
        campos=_f32c(campos, device, "campos"),
    )
    p = GsrParams()
    p.P = means3D.shape[0]
    p.D = int(degree)
    p.M = int(sh.shape[1]) if sh.numel() != 0 and sh.shape[0] != 0 else 0  # rasterize_points.cu:83-87
    p.W, p.H = int(W), int(H)
    p.tanfovx, p.tanfovy, p.scale_modifier = float(tan_fovx), float(tan_fovy), float(scale_modifier)
    p.prefiltered, p.debug, p.need_backward = int(bool(prefiltered)), int(bool(debug)), int(bool(need_backward))
    for k, t in keep.items():
        setattr(p, k, _ptr(t))
    return p, keep


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, need_backward=True):
    """Counterpart of RasterizeGaussiansCUDA (rasterize_points.cu:35-115); same argument order, same
    6-tuple result (num_rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer)."""
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    device = means3D.device
    _require_hip(device)
    P, H, W = means3D.shape[0], int(image_height), int(image_width)
    byte = dict(dtype=torch.uint8, device=device)
    if P == 0:  # rasterize_points.cu:81: the zero image (not the background) is returned
        e = torch.empty((0,), **byte)
        return (0, torch.zeros((3, H, W), dtype=torch.float32, device=device), torch.zeros((0,), dtype=torch.int32, device=device),
                e, e.clone(), e.clone())
    # every pixel and every radius is written by the kernels (the reference fills both with zeros first)
    out_color = torch.empty((3, H, W), dtype=torch.float32, device=device)
    radii = torch.empty((P,), dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        p, keep = _params(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                          viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, debug,
                          need_backward)
        geom = torch.empty((lib.gsr_geom_bytes(P),), **byte)
        img = torch.empty((lib.gsr_image_bytes(W, H),), **byte)
        R = C.c_int64(0)
        _check(lib.gsr_forward_stage1(C.byref(p), geom.data_ptr(), geom.numel(), img.data_ptr(), img.numel(),
                                      radii.data_ptr(), C.byref(R), stream))
        binning = torch.empty((lib.gsr_binning_bytes(R.value),), **byte)
        _check(lib.gsr_forward_stage2(C.byref(p), geom.data_ptr(), geom.numel(), binning.data_ptr(), binning.numel(),
                                      img.data_ptr(), img.numel(), R.value, out_color.data_ptr(), stream))
    del keep
    return int(R.value), out_color, radii, geom, binning, img


def recolor(background, means3D, colors, sh, degree, campos, image_height, image_width, num_rendered, geomBuffer,
            binningBuffer, imgBuffer, debug=False):
    """Re-render a finished forward's view with other per-Gaussian colours (exactly one of `colors` [P,3] / `sh`
    [P,M,3] non-empty), reusing its geometry, sorted lists and ranges (gsr_forward_recolor).  Returns color [3,H,W]."""
    device = means3D.device
    _require_hip(device)
    P, H, W = means3D.shape[0], int(image_height), int(image_width)
    out_color = torch.zeros((3, H, W), dtype=torch.float32, device=device)
    if P == 0:
        return out_color
    with torch.cuda.device(device):
        e = torch.empty(0)
        p, keep = _params(background, means3D, colors, torch.empty((1,), device=device), e, e, 1.0, e,
                          torch.empty((1,), device=device), torch.empty((1,), device=device), 1.0, 1.0, H, W, sh, degree,
                          campos, False, debug, False)
        _check(lib.gsr_forward_recolor(C.byref(p), geomBuffer.data_ptr(), geomBuffer.numel(), binningBuffer.data_ptr(),
                                       binningBuffer.numel(), imgBuffer.data_ptr(), imgBuffer.numel(), int(num_rendered),
                                       out_color.data_ptr(), torch.cuda.current_stream(device).cuda_stream))
        del keep
    return out_color


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos,
                                 geomBuffer, R, binningBuffer, imageBuffer, debug):
    """Counterpart of RasterizeGaussiansBackwardCUDA (rasterize_points.cu:117-196); same argument order,
    same 8-tuple (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)."""
    device = means3D.device
    _require_hip(device)
    P = means3D.shape[0]
    H, W = int(dL_dout_color.shape[1]), int(dL_dout_color.shape[2])
    M = int(sh.shape[1]) if sh.numel() != 0 and sh.shape[0] != 0 else 0
    z = dict(dtype=torch.float32, device=device)
    # Cleared by the caller (include/gsr.h): the [P,16] accumulation records of the render backward and dL_dsh (unused
    # rows stay zero) -- slices of ONE zero-filled allocation, so the clearing is one fill kernel.  Everything else is
    # written for every Gaussian by the per-Gaussian backward kernel.
    has_sr = scales.numel() != 0 and P != 0
    n_rec = 16 * P
    flat = torch.zeros((n_rec + 3 * M * P,), **z)
    grad_rec = flat[:n_rec]
    dL_dsh = flat[n_rec:].view(P, M, 3)
    e_or_z = torch.empty if P != 0 else torch.zeros
    dL_dmeans2D = e_or_z((P, 3), **z)
    dL_dcolors = e_or_z((P, 3), **z)
    dL_dopacity = e_or_z((P, 1), **z)
    dL_dmeans3D = e_or_z((P, 3), **z)
    dL_dcov3D = e_or_z((P, 6), **z)
    dL_dscales = torch.empty((P, 3), **z) if has_sr else torch.zeros((P, 3), **z)
    dL_drotations = torch.empty((P, 4), **z) if has_sr else torch.zeros((P, 4), **z)
    if P != 0:
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            opacity_unused = torch.empty((1,), **z)  # opacity lives in the geom arena; pointer only has to be non-NULL
            p, keep = _params(background, means3D, colors, opacity_unused, scales, rotations, scale_modifier,
                              cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, False,
                              debug, True)
            dpix = _f32c(dL_dout_color, device, "dL_dout_color")
            radii_c = radii.contiguous()
            _check(lib.gsr_backward(C.byref(p), radii_c.data_ptr(), int(R), geomBuffer.data_ptr(), geomBuffer.numel(),
                                    binningBuffer.data_ptr(), binningBuffer.numel(), imageBuffer.data_ptr(),
                                    imageBuffer.numel(), dpix.data_ptr(), dL_dmeans2D.data_ptr(), grad_rec.data_ptr(),
                                    dL_dopacity.data_ptr(), dL_dcolors.data_ptr(), dL_dmeans3D.data_ptr(),
                                    dL_dcov3D.data_ptr(), _ptr(dL_dsh), dL_dscales.data_ptr(), dL_drotations.data_ptr(),
                                    stream))
            del keep
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def mark_visible(means3D, viewmatrix, projmatrix):
    """Counterpart of markVisible (rasterize_points.cu:198-217)."""
    device = means3D.device
    _require_hip(device)
    P = means3D.shape[0]
    present = torch.zeros((P,), dtype=torch.bool, device=device)
    if P != 0:
        with torch.cuda.device(device):
            m = _f32c(means3D, device, "means3D")
            v = _f32c(viewmatrix, device, "viewmatrix")
            pr = _f32c(projmatrix, device, "projmatrix")
            _check(lib.gsr_mark_visible(P, m.data_ptr(), v.data_ptr(), pr.data_ptr(), present.data_ptr(),
                                        torch.cuda.current_stream(device).cuda_stream))
    return present


# ---- inspection helpers for tests / bench (not part of the reference API) -------------------------------
_QSPEC = {
    "DEPTHS": (torch.float32, lambda P, R, T, N: (P,)), "MEANS2D": (torch.float32, lambda P, R, T, N: (P, 2)),
    "CONIC_OPACITY": (torch.float32, lambda P, R, T, N: (P, 4)), "RGB": (torch.float32, lambda P, R, T, N: (P, 3)),
    "TILES_TOUCHED": (torch.int32, lambda P, R, T, N: (P,)), "POINT_LIST": (torch.int32, lambda P, R, T, N: (R,)),
    "POINT_LIST_KEYS": (torch.int64, lambda P, R, T, N: (R,)), "RANGES": (torch.int32, lambda P, R, T, N: (T, 2)),
    "FINAL_T": (torch.float32, lambda P, R, T, N: (N,)), "N_CONTRIB": (torch.int32, lambda P, R, T, N: (N,)),
    "CLAMPED": (torch.uint8, lambda P, R, T, N: (P, 3)), "TILE_NEED": (torch.int32, lambda P, R, T, N: (T,)),
}


def query(name, P, W, H, R, geom, binning, img):
    """Copy one private arena array out (device tensor).  Unsigned data come back in same-width signed dtypes."""
    dtype, shp = _QSPEC[name]
    T = ((W + 15) // 16) * ((H + 15) // 16)
    out = torch.zeros(shp(P, R, T, W * H), dtype=dtype, device=geom.device)
    if out.numel() == 0:
        return out
    p = GsrParams()
    p.P, p.W, p.H = P, W, H
    with torch.cuda.device(geom.device):
        _check(lib.gsr_query(C.byref(p), Q[name], geom.data_ptr(), _ptr(binning), img.data_ptr(), int(R), out.data_ptr(),
                             out.numel() * out.element_size(), torch.cuda.current_stream(geom.device).cuda_stream))
    return out


def selftest(device):
    with torch.cuda.device(device):
        _check(lib.gsr_selftest(torch.cuda.current_stream(device).cuda_stream))


def set_profiling(on):
    lib.gsr_set_profiling(int(bool(on)))


def get_profile():
    names = (C.c_char_p * 256)()
    ms = (C.c_float * 256)()
    n = lib.gsr_get_profile(names, ms, 256)
    return [(names[i].decode(), float(ms[i])) for i in range(n)]
