"""diff_gaussian_rasterization -- MI355X-native drop-in for the reference package of the same name.

Public surface mirrors /root/reference/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py
field-for-field so the reference's callers (simple_raw_render.py:12,98-111,263-277) run unchanged on
PyTorch-ROCm:

    GaussianRasterizationSettings   NamedTuple, 12 fields, same order        (__init__.py:157-169)
    GaussianRasterizer(nn.Module)   .forward(means3D, means2D, opacities, shs=, colors_precomp=, scales=,
                                    rotations=, cov3D_precomp=) -> (color[3,H,W], radii[P]); .markVisible
                                                                                                (:171-220)
    rasterize_gaussians, _RasterizeGaussians (autograd Function, 9 inputs, grads in input order) (:21-155)
    cpu_deep_copy_tuple                                                                         (:17-19)

Underneath, ``_native`` (ctypes over the C-ABI HIP library, include/gsr.h) takes the place of the pybind
module ``_C``.  No CPU path exists: tensors must live on a HIP device.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _native as _C


def cpu_deep_copy_tuple(input_tuple):
    """Host copies of every tensor of an argument tuple (what the debug snapshots store); other items pass through."""
    return tuple(x.cpu().clone() if isinstance(x, torch.Tensor) else x for x in input_tuple)


def _native_call(fn, args, debug, dump, message, **kw):
    """One call into the native library.  With `debug` the arguments are copied to the host first, and if the call raises they are
    written to `dump` with the reference's message before the exception travels on (reference __init__.py:83-90,132-139)."""
    if not debug:
        return fn(*args, **kw)
    snapshot = cpu_deep_copy_tuple(args)
    try:
        return fn(*args, **kw)
    except Exception:
        torch.save(snapshot, dump)
        print(message)
        raise


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings)


def _or_none(grad, inp):
    """an absent optional was passed as an empty tensor: its gradient slot gets None, not a [P, ...] tensor of the wrong shape
    (autograd validates shapes on ROCm torch 2.x)"""
    return grad if inp.numel() != 0 else None


class _RasterizeGaussians(torch.autograd.Function):
    """The per-view call: 9 inputs in the reference's order, (color, radii) out, gradients back in input order (reference
    __init__.py:44-155)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        rs = raster_settings
        if not rs.debug:
            # steady state of a per-view loop: the short path (_native.forward_view; None = take the general one below)
            need_backward = any(ctx.needs_input_grad)
            fast = _C.forward_view(rs, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, need_backward)
            if fast is not None:
                num_rendered, color, radii, arenas, layout = fast
                ctx.raster_settings, ctx.num_rendered, ctx.opacity_shape, ctx.layout = rs, num_rendered, tuple(opacities.shape), layout
                if need_backward:
                    ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, arenas)
                ctx.mark_non_differentiable(radii)
                return color, radii
        ctx.layout = None
        # the native call's argument order = reference __init__.py:60-80
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered,
                rs.debug)
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _native_call(
            _C.rasterize_gaussians, args, rs.debug, "snapshot_fw.dump",
            "\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.", need_backward=any(ctx.needs_input_grad))
        ctx.raster_settings, ctx.num_rendered, ctx.opacity_shape = rs, num_rendered, tuple(opacities.shape)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
                              imgBuffer)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        rs = ctx.raster_settings
        if ctx.layout is not None:
            colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, arenas = ctx.saved_tensors
            (g_means2D, g_colors, g_opacities, g_means3D, g_cov3D, g_sh, g_scales, g_rotations) = _C.backward_view(
                rs, means3D, radii, colors_precomp, scales, rotations, cov3Ds_precomp, grad_out_color, sh, arenas, ctx.layout)
            return (g_means3D, g_means2D, _or_none(g_sh, sh), _or_none(g_colors, colors_precomp), g_opacities.reshape(ctx.opacity_shape),
                    _or_none(g_scales, scales), _or_none(g_rotations, rotations), _or_none(g_cov3D, cov3Ds_precomp), None)
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
         imgBuffer) = ctx.saved_tensors
        # the native call's argument order = reference __init__.py:109-129
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh, rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered,
                binningBuffer, imgBuffer, rs.debug)
        (g_means2D, g_colors, g_opacities, g_means3D, g_cov3D, g_sh, g_scales, g_rotations) = _native_call(
            _C.rasterize_gaussians_backward, args, rs.debug, "snapshot_bw.dump",
            "\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
        # gradients in the order of forward's inputs; the settings get none
        return (g_means3D, g_means2D, _or_none(g_sh, sh), _or_none(g_colors, colors_precomp), g_opacities.reshape(ctx.opacity_shape),
                _or_none(g_scales, scales), _or_none(g_rotations, rotations), _or_none(g_cov3D, cov3Ds_precomp), None)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


_ABSENT = torch.Tensor([])


class GaussianRasterizer(nn.Module):
    """nn.Module like the reference's (no parameters, no buffers: the settings are its only state).  The reference's caller builds one
    PER CALL (simple_raw_render.py:263), between the copies that drain the stream and the first kernel, so construction and call are
    kept off nn.Module's bookkeeping until something asks for it: nn.Module.__init__ (a dozen dictionaries) runs on the first access
    to module state -- hooks, .to(), state_dict(), children ... -- and a module nobody has touched that way is called straight
    through to forward (there can be no hooks on it)."""

    def __init__(self, raster_settings):
        object.__setattr__(self, "raster_settings", raster_settings)

    def __getattr__(self, name):
        if "_parameters" not in self.__dict__:          # nn.Module state asked for: build it now
            rs = self.__dict__.pop("raster_settings")
            nn.Module.__init__(self)
            object.__setattr__(self, "raster_settings", rs)
            return getattr(self, name)
        return nn.Module.__getattr__(self, name)

    def __setattr__(self, name, value):
        if "_parameters" not in self.__dict__:
            self.__getattr__("_parameters")
        nn.Module.__setattr__(self, name, value)

    def __call__(self, *args, **kwargs):
        if "_parameters" not in self.__dict__:
            return self.forward(*args, **kwargs)
        return nn.Module.__call__(self, *args, **kwargs)

    def markVisible(self, positions):
        with torch.no_grad():
            raster_settings = self.raster_settings
            visible = _C.mark_visible(positions, raster_settings.viewmatrix, raster_settings.projmatrix)
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        # exactly one colour source and exactly one covariance source (reference __init__.py:190-195: the same exception texts)
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        pair_given, pair_complete = scales is not None or rotations is not None, scales is not None and rotations is not None
        if (cov3D_precomp is None and not pair_complete) or (cov3D_precomp is not None and pair_given):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        # absent optionals travel as empty CPU tensors, like the reference's torch.Tensor([]) (:197-206); one shared empty tensor
        # instead of a fresh one per call (this method runs while the GPU waits for the call's first kernel)
        absent = _ABSENT
        shs, colors_precomp, scales, rotations, cov3D_precomp = (absent if t is None else t
                                                                 for t in (shs, colors_precomp, scales, rotations, cov3D_precomp))
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)


# ---- extension: all views of one cloud in ONE submission ----------------------------------------------------
# The reference's caller loops views in Python, one GaussianRasterizer call each (simple_raw_render.py:259-278).  The entry
# points below take the list of per-view settings instead and hand the whole batch to the C ABI's gsr_forward_batch /
# gsr_backward_batch: one preprocess grid over V x P, one render grid over all views' tiles, and gradients summed over the
# views on the device -- the same numbers as V separate calls whose gradients autograd adds up.
# The per-view matrices of a settings list are packed ONCE into [V,4,4] / [V,4,4] / [V,3] device blocks and remembered:
# a training loop hands over the same settings objects every iteration, and rebuilding the blocks (36 tiny copies + 3 stacks)
# or comparing backgrounds through host memory (a blocking device->host copy per view, which drains the stream) would put
# the host back on the critical path of every call.  Steady state: no torch kernel, no copy, no synchronisation.
#
# What the cache can and cannot see.  A tensor is identified by (address, shape, strides, version counter, device): in-place
# edits through torch (`m.copy_(..)`, `m[0, 0] = ..`, `m.mul_(..)`) bump the version counter and repack the blocks.  Writes
# that bypass the counter -- `m.data.copy_(..)`, a numpy array sharing a CPU tensor's memory -- are invisible to it and would
# render with the packed (stale) matrices: build new tensors for new cameras, or switch the cache off
# (GSR_VIEW_CACHE=0 in the environment, or diff_gaussian_rasterization.set_view_cache(False)): the blocks are then packed
# on every call, as the reference's caller does.  Tensors created under torch.inference_mode() carry no version counter: a
# settings list that contains one is never cached.
import os as _os

_VIEW_BLOCKS = {}          # key -> (view, proj, cam, the source tensors kept alive so that their addresses cannot be reused)
_VIEW_BLOCKS_MAX = 32
_BG_SAME = {}              # (key of a, key of b) -> the two background tensors hold the same three values
_VIEW_CACHE_ON = _os.environ.get("GSR_VIEW_CACHE", "1") != "0"


def set_view_cache(on):
    """Switch the packed-view-block cache of rasterize_views on or off (off: pack the per-view matrices on every call)."""
    global _VIEW_CACHE_ON
    _VIEW_CACHE_ON = bool(on)
    if not on:
        _VIEW_BLOCKS.clear()
        _BG_SAME.clear()


def _tkey(t):
    """Identity of a tensor's current contents as far as torch tracks it, or None when it does not (inference tensors)."""
    try:
        ver = t._version
    except RuntimeError:
        return None
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), ver, t.device.type, t.device.index)


def _same_background(a, b):
    if a is b or (a.data_ptr() == b.data_ptr() and a.device == b.device and a.shape == b.shape and a.stride() == b.stride()):
        return True
    ka, kb = _tkey(a), _tkey(b)
    if not _VIEW_CACHE_ON or ka is None or kb is None:
        return torch.equal(a.detach().cpu(), b.detach().cpu())
    k = ka + kb
    r = _BG_SAME.get(k)
    if r is None:
        if len(_BG_SAME) >= 256:
            _BG_SAME.clear()
        r = (torch.equal(a.detach().cpu(), b.detach().cpu()), a, b)     # one host comparison per pair of tensors, ever
        _BG_SAME[k] = r
    return r[0]


def _pack_views(settings_list, device):
    view = torch.stack([s.viewmatrix.reshape(4, 4).to(device) for s in settings_list], 0).contiguous()
    proj = torch.stack([s.projmatrix.reshape(4, 4).to(device) for s in settings_list], 0).contiguous()
    cam = torch.stack([s.campos.reshape(3).to(device) for s in settings_list], 0).contiguous()
    return view, proj, cam


def _stack_views(settings_list, device):
    s0 = settings_list[0]
    for s in settings_list[1:]:
        if (s.image_height, s.image_width, s.tanfovx, s.tanfovy, s.scale_modifier, s.sh_degree, s.prefiltered) != (
                s0.image_height, s0.image_width, s0.tanfovx, s0.tanfovy, s0.scale_modifier, s0.sh_degree, s0.prefiltered) or \
                not _same_background(s.bg, s0.bg):
            raise Exception("rasterize_views: the views of a batch must share image size, tan(fov), background, scale "
                            "modifier, SH degree and the prefiltered flag")
    _C._require_hip(device)
    keys = [_tkey(t) for s in settings_list for t in (s.viewmatrix, s.projmatrix, s.campos)]
    if not _VIEW_CACHE_ON or any(k is None for k in keys):
        return _pack_views(settings_list, device)
    key = (device.type, device.index) + tuple(keys)
    hit = _VIEW_BLOCKS.get(key)
    cur = torch.cuda.current_stream(device)
    if hit is None:
        view, proj, cam = _pack_views(settings_list, device)
        # other host threads may pick the blocks up on other streams: they wait (on the device) for this event, nobody waits
        # on the host
        ev = torch.cuda.Event()
        ev.record(cur)
        if len(_VIEW_BLOCKS) >= _VIEW_BLOCKS_MAX:
            _VIEW_BLOCKS.pop(next(iter(_VIEW_BLOCKS)))
        hit = (view, proj, cam, [(s.viewmatrix, s.projmatrix, s.campos) for s in settings_list], ev, cur.cuda_stream)
        _VIEW_BLOCKS[key] = hit
    elif hit[5] != cur.cuda_stream:
        cur.wait_event(hit[4])
    return hit[0], hit[1], hit[2]


class _RasterizeGaussiansViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings_list):
        rs = settings_list[0]
        view, proj, cam = _stack_views(settings_list, means3D.device)
        need_backward = any(ctx.needs_input_grad)
        counts, color, radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians_batch(
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, view, proj,
            rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree, cam, rs.prefiltered,
            any(s.debug for s in settings_list), need_backward=need_backward)
        ctx.raster_settings = rs
        ctx.num_rendered = counts
        ctx.opacity_shape = tuple(opacities.shape)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
                              imgBuffer, view, proj, cam)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer, view, proj,
         cam) = ctx.saved_tensors
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = _C.rasterize_gaussians_backward_batch(
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, view, proj, rs.tanfovx,
            rs.tanfovy, grad_out_color, sh, rs.sh_degree, cam, geomBuffer, binningBuffer, imgBuffer, rs.debug)

        def fit(g, inp):
            return g if inp.numel() != 0 else None

        return (grad_means3D, grad_means2D, fit(grad_sh, sh), fit(grad_colors_precomp, colors_precomp),
                grad_opacities.reshape(ctx.opacity_shape), fit(grad_scales, scales), fit(grad_rotations, rotations),
                fit(grad_cov3Ds_precomp, cov3Ds_precomp), None)


def rasterize_views(means3D, means2D, opacities, settings_list, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None):
    """GaussianRasterizer.forward for a LIST of GaussianRasterizationSettings (views of one cloud) in one call.
    Returns (colors [V,3,H,W], radii [V,P]); gradients of the shared inputs are the sums over the views."""
    if len(settings_list) == 0:
        raise Exception("rasterize_views: empty settings list")
    if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    if ((scales is None or rotations is None) and cov3D_precomp is None) or (
            (scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
    e = _ABSENT
    return _RasterizeGaussiansViews.apply(
        means3D, means2D, e if shs is None else shs, e if colors_precomp is None else colors_precomp, opacities,
        e if scales is None else scales, e if rotations is None else rotations, e if cov3D_precomp is None else cov3D_precomp,
        list(settings_list))
