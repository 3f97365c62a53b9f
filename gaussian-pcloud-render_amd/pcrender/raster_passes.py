"""Caller-side glue of the reference's render path, on top of the drop-in rasterizer.

Mirrors, for inference use (the reference calls these under torch.no_grad(), simple_benchmark.py:198):
  PCML_Render._rasterize   /root/reference/simple_raw_render.py:227-288   -> rasterize_views
  the four passes of PCML_Render.render, :410-524 (world xyz, SH colour, hit map, normals) -> render_passes

`rasterize_views` is the literal call pattern: one GaussianRasterizer call per (batch item, view), stack,
bilinear down-filter when super_sample_rate > 1, permute to (b, q, h, w, 3).

`render_passes` produces the same four images per view but runs the geometry (preprocess, depth sort, pair
emission, tile sort, ranges) ONCE per view -- all views in one submission (C ABI gsr_forward_batch) -- and re-renders the
other colours on it with diff_gaussian_rasterization._native.recolor (C ABI gsr_forward_recolor), again all views per
call; every pass is bit-identical to the corresponding full call.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from diff_gaussian_rasterization import rasterize_views as _rasterize_views_call
from diff_gaussian_rasterization import _native

from . import camera as _camera


def settings_for_view(H_c2w, width_px, height_px, fov_deg, device, sh_degree=0, bg=None, super_sample_rate=2):
    """get_rasterize_param_from_camera for one view (H_c2w [4,4], CPU or device)."""
    a = _camera.raster_settings_arrays(H_c2w.detach().cpu().float(), width_px, height_px, fov_deg, super_sample_rate)
    bg = torch.zeros(3, device=device) if bg is None else bg.to(device)
    return GaussianRasterizationSettings(
        image_height=a["image_height"], image_width=a["image_width"], tanfovx=a["tanfovx"], tanfovy=a["tanfovy"], bg=bg,
        scale_modifier=1.0, viewmatrix=a["viewmatrix"].to(device), projmatrix=a["projmatrix"].to(device),
        sh_degree=sh_degree, campos=a["campos"].to(device), prefiltered=False, debug=False)


def _finish(frames, batchsize, num_q, h, w, ss):
    """stack -> [b*q,3,h*ss,w*ss] -> bilinear down-filter -> (b, q, h, w, 3)   (simple_raw_render.py:279-288)"""
    x = torch.stack(frames, dim=0).reshape(batchsize * num_q, 3, h * ss, w * ss)
    if ss > 1:
        x = F.interpolate(x, size=(h, w), mode="bilinear", align_corners=False)
    return x.reshape(batchsize, num_q, 3, h, w).permute(0, 1, 3, 4, 2)


def rasterize_views(means3D_list, opacity_list, scales_list, rotations_list, H_c2w, h, w, fov, bg, scale_factor,
                    shs_list=None, colors_list=None, sh_degree=1, super_sample_rate=2, normalize_camera_normal=False,
                    batch_views=False):
    """The reference's _rasterize: lists are per batch item, H_c2w is [b, q, 4, 4] (Camera.H_c2w).  batch_views=True submits
    the q views of a batch item in ONE rasterizer call (diff_gaussian_rasterization.rasterize_views; same images, gradients
    summed over the views like autograd does for the loop) whenever the colours do not depend on the view."""
    batchsize, num_q = H_c2w.shape[0], H_c2w.shape[1]
    frames = []
    for i in range(batchsize):
        means3D = means3D_list[i]
        device = means3D.device
        means2D = torch.zeros_like(means3D, dtype=torch.float32, requires_grad=True, device=device) + 0
        radius = float(np.sqrt(3) / scale_factor * 6)   # simple_raw_render.py:248
        scales = scales_list[i] * radius
        colors_i = None if colors_list is None else colors_list[i]
        if batch_views and not normalize_camera_normal:
            sts = [settings_for_view(H_c2w[i, j], w, h, fov, device, sh_degree=sh_degree, bg=bg, super_sample_rate=super_sample_rate)
                   for j in range(num_q)]
            imgs, _ = _rasterize_views_call(means3D, means2D, opacity_list[i], sts, shs=None if shs_list is None else shs_list[i],
                                            colors_precomp=colors_i, scales=scales, rotations=rotations_list[i])
            frames.extend(list(imgs))
            continue
        for j in range(num_q):
            st = settings_for_view(H_c2w[i, j], w, h, fov, device, sh_degree=sh_degree, bg=bg, super_sample_rate=super_sample_rate)
            if normalize_camera_normal:                 # simple_raw_render.py:264-268, incl. the sign-of-first-point quirk (Q11)
                cam_orig = H_c2w[i, j, :3, 3].to(device)
                sgn = (torch.sum((means3D - cam_orig) * colors_i, -1, keepdim=True) > 0).float() * 2 - 1
                colors_i = colors_i * (-1) * sgn[0]
            img, _ = GaussianRasterizer(st)(
                means3D=means3D, means2D=means2D, shs=None if shs_list is None else shs_list[i], colors_precomp=colors_i,
                opacities=opacity_list[i], scales=scales, rotations=rotations_list[i], cov3D_precomp=None)
            frames.append(img)
    return _finish(frames, batchsize, num_q, h, w, super_sample_rate)


@torch.no_grad()
def render_passes(means3D, opacities, scales, rotations, shs, H_c2w, h, w, fov, bg, scale_factor, normals=None, sh_degree=1,
                  super_sample_rate=2):
    """The four passes of PCML_Render.render for ONE cloud and q views (H_c2w [q,4,4]); returns a dict of
    (1, q, h, w, 3) tensors: 'rgb', 'xyz_w', 'hitmap' and 'normal' (None without normals).

    All q views go through the pipeline front end in ONE submission (gsr_forward_batch, with the SH colours); xyz, ones and
    the normals are re-rendered on the same sorted lists, again all views per call (the normals with per-view colours: the
    reference flips their sign view by view).  Inference only (no autograd graph), like the reference's use."""
    device = means3D.device
    num_q = H_c2w.shape[0]
    radius = float(np.sqrt(3) / scale_factor * 6)
    sc = (scales * radius).contiguous()
    e = torch.empty(0)
    sts = [settings_for_view(H_c2w[j], w, h, fov, device, sh_degree=sh_degree, bg=bg, super_sample_rate=super_sample_rate)
           for j in range(num_q)]
    st = sts[0]
    H, W = st.image_height, st.image_width
    view = torch.stack([s.viewmatrix.reshape(4, 4) for s in sts]).contiguous()
    proj = torch.stack([s.projmatrix.reshape(4, 4) for s in sts]).contiguous()
    cam = torch.stack([s.campos.reshape(3) for s in sts]).contiguous()
    counts, rgb, radii, geom, binning, img = _native.rasterize_gaussians_batch(
        st.bg, means3D, e, opacities, sc, rotations, 1.0, e, view, proj, st.tanfovx, st.tanfovy, H, W, shs, sh_degree, cam, False,
        False, need_backward=False)

    def again(colors):
        return _native.recolor(st.bg, means3D, colors, e, 0, cam, H, W, counts, geom, binning, img).reshape(num_q, 3, H, W)

    out = {"rgb": rgb, "xyz_w": again(means3D), "hitmap": again(torch.ones_like(means3D)), "normal": None}
    if normals is not None:
        per_view = []
        colors_n = normals
        for j in range(num_q):
            cam_orig = H_c2w[j, :3, 3].to(device)
            sgn = (torch.sum((means3D - cam_orig) * colors_n, -1, keepdim=True) > 0).float() * 2 - 1
            colors_n = colors_n * (-1) * sgn[0]          # carried across views exactly like the reference's loop
            per_view.append(colors_n)
        out["normal"] = again(torch.stack(per_view, 0).contiguous())
    return {k: (None if v is None else _finish(list(v), 1, num_q, h, w, super_sample_rate)) for k, v in out.items()}
