"""Caller-side glue of the reference's render path, on top of the drop-in rasterizer.

Mirrors, for inference use (the reference calls these under torch.no_grad(), simple_benchmark.py:198):
  PCML_Render._rasterize   /root/reference/simple_raw_render.py:227-288   -> rasterize_views
  the four passes of PCML_Render.render, :410-524 (world xyz, SH colour, hit map, normals) -> render_passes

`rasterize_views` is the literal call pattern: one GaussianRasterizer call per (batch item, view), stack,
bilinear down-filter when super_sample_rate > 1, permute to (b, q, h, w, 3).

`literal_passes` is the reference's pass sequence as it stands: one `rasterize_views` per pass, in PCML_Render.render's order (world
xyz, SH colour, hit map, normals; :387-524) or Simple_Render.render's (:740-820), pinned call by call -- every argument of every
rasterizer call and the post-processed results -- by tests/golden/py_rasterize_calls.npz, the trace of the reference's own code.

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


def settings_for_views(H_c2w, width_px, height_px, fov_deg, device, sh_degree=0, bg=None, super_sample_rate=2):
    """settings_for_view for q views (H_c2w [q,4,4]): the matrices of all views are computed in one go on the host and reach
    the device in one copy; the per-view settings hold slices of it (same values bit for bit)."""
    q = H_c2w.shape[0]
    a = _camera.raster_settings_batch(H_c2w.detach().cpu().float(), width_px, height_px, fov_deg, super_sample_rate)
    packed = torch.cat([a["viewmatrix"].reshape(q, 16), a["projmatrix"].reshape(q, 16), a["campos"]], dim=1).to(device)
    bg = torch.zeros(3, device=device) if bg is None else bg.to(device)
    return [GaussianRasterizationSettings(
        image_height=a["image_height"], image_width=a["image_width"], tanfovx=a["tanfovx"], tanfovy=a["tanfovy"], bg=bg,
        scale_modifier=1.0, viewmatrix=packed[j, :16].reshape(4, 4), projmatrix=packed[j, 16:32].reshape(4, 4),
        sh_degree=sh_degree, campos=packed[j, 32:35], prefiltered=False, debug=False) for j in range(q)]


def _finish(frames, batchsize, num_q, h, w, ss):
    """stack -> [b*q,3,h*ss,w*ss] -> bilinear down-filter -> (b, q, h, w, 3)   (simple_raw_render.py:279-288)"""
    x = frames if torch.is_tensor(frames) else torch.stack(frames, dim=0)     # a [b*q,3,H,W] tensor is taken as it is
    x = x.reshape(batchsize * num_q, 3, h * ss, w * ss)
    if ss > 1:
        x = F.interpolate(x, size=(h, w), mode="bilinear", align_corners=False)
    return x.reshape(batchsize, num_q, 3, h, w).permute(0, 1, 3, 4, 2)


def rasterize_views(means3D_list, opacity_list, scales_list, rotations_list, H_c2w, h, w, fov, bg, scale_factor,
                    shs_list=None, colors_list=None, sh_degree=1, super_sample_rate=2, normalize_camera_normal=False,
                    batch_views=False, simple=False):
    """The reference's _rasterize: lists are per batch item, H_c2w is [b, q, 4, 4] (Camera.H_c2w).  batch_views=True submits
    the q views of a batch item in ONE rasterizer call (diff_gaussian_rasterization.rasterize_views; same images, gradients
    summed over the views like autograd does for the loop) whenever the colours do not depend on the view.
    simple=False: PCML_Render._rasterize (simple_raw_render.py:227-288), which scales the decoded scales by
    sqrt(3) / scale_factor * 6; simple=True: Simple_Render._rasterize (:599-660), which takes the scales as they are (the factor
    is commented out there, :617) and renders with opacity 1 whatever it is given (:616)."""
    batchsize, num_q = H_c2w.shape[0], H_c2w.shape[1]
    frames = []
    for i in range(batchsize):
        means3D = means3D_list[i]
        device = means3D.device
        means2D = torch.zeros_like(means3D, dtype=torch.float32, requires_grad=True, device=device) + 0
        if simple:
            scales, opacity_i = scales_list[i], torch.ones_like(opacity_list[i])
        else:
            radius = float(np.sqrt(3) / scale_factor * 6)   # simple_raw_render.py:248
            scales, opacity_i = scales_list[i] * radius, opacity_list[i]
        colors_i = None if colors_list is None else colors_list[i]
        sts = settings_for_views(H_c2w[i], w, h, fov, device, sh_degree=sh_degree, bg=bg, super_sample_rate=super_sample_rate)
        if batch_views and not normalize_camera_normal:
            imgs, _ = _rasterize_views_call(means3D, means2D, opacity_i, sts, shs=None if shs_list is None else shs_list[i],
                                            colors_precomp=colors_i, scales=scales, rotations=rotations_list[i])
            frames.extend(list(imgs))
            continue
        for j in range(num_q):
            st = sts[j]
            if normalize_camera_normal:
                # simple_raw_render.py:264-268: every normal is turned towards the camera of the view being rendered, and the
                # turned array is carried into the next view.  (camera_orig is [1,1,3] there, so the sign tensor is [1,N,1] and its
                # `[0]` strips the BATCH axis: the signs are per point -- the call trace of the reference's own loop,
                # tests/golden/py_rasterize_calls.npz, shows 24 different sign patterns; SURVEY.md's quirk Q11 misread it.)
                cam_orig = H_c2w[i, j, :3, 3].to(device)
                sgn = (torch.sum((means3D - cam_orig) * colors_i, -1, keepdim=True) > 0).float() * 2 - 1
                colors_i = colors_i * (-1) * sgn
            img, _ = GaussianRasterizer(st)(
                means3D=means3D, means2D=means2D, shs=None if shs_list is None else shs_list[i], colors_precomp=colors_i,
                opacities=opacity_i, scales=scales, rotations=rotations_list[i], cov3D_precomp=None)
            frames.append(img)
    return _finish(frames, batchsize, num_q, h, w, super_sample_rate)


def pcgc_rescale(xyz, offset=512, factor=256):
    """voxel coordinates -> world units in float32 torch arithmetic (simple_raw_render.py:73-77)"""
    return (xyz - offset) / factor


def simple_primitives(points, colors, sigma=1., scale_factor=1., voxelized=True, offset=512):
    """Simple_Render's per-Gaussian inputs from positions and colours in [0, 1] (simple_raw_render.py:688-726), torch float32 like
    the reference: SH = [RGB2SH(colour) | 12 zero rows] (sh_deg 1 -> pseudo_sh_dim 12 -> M = 13), identity quaternions, isotropic
    scales sigma (/ scale_factor for voxelised input), opacity 1, means = pcgc_rescale(points) if voxelised."""
    dc = ((colors - 0.5) / 0.28209479177387814).unsqueeze(-2)                    # models/sh_utils.py:114-115
    shs = torch.cat([dc, torch.zeros((colors.shape[0], 12, 3), device=colors.device)], dim=1)
    quat = torch.tensor([1, 0, 0, 0], dtype=torch.float32, device=colors.device).expand(colors.shape[0], 4)
    norm = scale_factor if voxelized else 1.
    return dict(means3D=pcgc_rescale(points.float(), offset, scale_factor) if voxelized else points.float(), shs=shs, rotations=quat,
                scales=torch.ones_like(colors[:, 0:3]) * sigma / norm, opacities=torch.ones_like(colors[:, 0:1]))


def literal_passes(means3D, opacities, scales, rotations, shs, H_c2w, h, w, fov, background_color, scale_factor, normals=None,
                   sh_degree=1, super_sample_rate=2, simple=False):
    """The reference's passes, one full `rasterize_views` each: PCML_Render.render's xyz_w, rgb, hitmap, normal
    (simple_raw_render.py:387-524) or, simple=True, Simple_Render.render's rgb, xyz_w, hitmap (:740-820; scales taken as they are,
    opacity 1) for ONE cloud (the reference keeps batch item 0, :377-382) and the q views of H_c2w [q,4,4].  Returns the
    reference's ret_dict: (1, q, h, w, 3) tensors."""
    order = ("rgb", "xyz_w", "hitmap", "normal") if simple else ("xyz_w", "rgb", "hitmap", "normal")
    bg = torch.zeros(3, device=means3D.device) + background_color                # :398
    colours = {"xyz_w": means3D, "hitmap": torch.ones_like(means3D), "normal": normals}
    out = {"rgb": None, "normal": None, "xyz_w": None, "hitmap": None}
    for name in order:
        if name == "normal" and normals is None:
            continue
        out[name] = rasterize_views(
            [means3D], [opacities], [scales], [rotations], H_c2w.unsqueeze(0), h, w, fov, bg, scale_factor,
            shs_list=[shs] if name == "rgb" else None, colors_list=None if name == "rgb" else [colours[name]],
            sh_degree=sh_degree, super_sample_rate=super_sample_rate, normalize_camera_normal=(name == "normal"), simple=simple)
    return out


def normals_per_view(means3D, normals, cam_origins):
    """What the reference's per-view loop renders as normals in view j (simple_raw_render.py:264-268), for all views: [q,N,3].
    n_j = -n_(j-1) * sgn_j with sgn_j = +1 where (p - cam_j) . n_(j-1) > 0, else -1, per point, n_(-1) = the given normals: every
    normal ends up turned towards camera j; a normal at exactly 90 degrees keeps the direction the previous view left it with,
    which is why the views are walked in order like the reference does.  The same torch expressions as the loop, so the values are
    the loop's bit for bit (multiplying by +-1 is exact)."""
    cur, out = normals, []
    for j in range(cam_origins.shape[0]):
        sgn = (torch.sum((means3D - cam_origins[j]) * cur, -1, keepdim=True) > 0).float() * 2 - 1
        cur = cur * (-1) * sgn
        out.append(cur)
    return torch.stack(out, 0)


@torch.no_grad()
def render_passes(means3D, opacities, scales, rotations, shs, H_c2w, h, w, fov, bg, scale_factor, normals=None, sh_degree=1,
                  super_sample_rate=2):
    """The four passes of PCML_Render.render for ONE cloud and q views (H_c2w [q,4,4]); returns a dict of
    (1, q, h, w, 3) tensors: 'rgb', 'xyz_w', 'hitmap' and 'normal' (None without normals).

    All q views go through the pipeline in ONE submission, and the four passes in ONE render: world xyz, the hit map and the
    normals are extra channels of the colour render (gsr_forward_batch_channels; the normals as one value array per view: the
    reference turns every normal towards the camera of the view it renders).  With a background whose three values differ the
    hit map's channels differ too, and the passes are re-rendered one by one on the sorted lists instead (gsr_forward_recolor).
    Inference only (no autograd graph), like the reference's use."""
    device = means3D.device
    num_q = H_c2w.shape[0]
    radius = float(np.sqrt(3) / scale_factor * 6)
    sc = (scales * radius).contiguous()
    e = torch.empty(0)
    # the settings of all views in one go, one host -> device copy (get_rasterize_param_from_camera per view is 0.2 ms of host
    # time each: more than the views' whole front end takes on the GPU)
    H_cpu = H_c2w.detach().cpu().float()
    a = _camera.raster_settings_batch(H_cpu, w, h, fov, super_sample_rate)
    H, W = a["image_height"], a["image_width"]
    packed = torch.cat([a["viewmatrix"].reshape(num_q, 16), a["projmatrix"].reshape(num_q, 16), a["campos"],
                        H_cpu[:, :3, 3]], dim=1).to(device)
    view, proj, cam = packed[:, :16].contiguous(), packed[:, 16:32].contiguous(), packed[:, 32:35].contiguous()
    bg_d = torch.zeros(3, device=device) if bg is None else bg.to(device)
    nv = None if normals is None else normals_per_view(means3D, normals, packed[:, 35:38])        # [q, P, 3]
    bg_cpu = bg if (bg is not None and bg.device.type == "cpu") else bg_d.cpu()
    if float(bg_cpu[0]) == float(bg_cpu[1]) == float(bg_cpu[2]):
        # ONE render for the four passes: world xyz, the hit map's single channel and the normals ride along as extra channels
        # of the colour render (gsr_forward_batch_channels) -- same alphas, same stopping decisions, the same sums term for term
        P = means3D.shape[0]
        one = torch.ones((P, 1), dtype=torch.float32, device=device)
        extra = torch.cat([means3D, one], dim=1).contiguous()                                               # [P, 4], shared by the views
        nx = 4
        if nv is not None:
            # + the turned normals, one [P, 4] array per view; xyz and the hit value stay shared (split layout: 16 B per point
            # + 16 B per point and view instead of 32 B per point and view)
            extra = (extra, torch.cat([nv, torch.zeros((num_q, P, 1), dtype=torch.float32, device=device)], dim=2).contiguous())
            nx = 8
        counts, rgb, radii, geom, binning, img, ex = _native.rasterize_gaussians_batch(
            bg_d, means3D, e, opacities, sc, rotations, 1.0, e, view, proj, a["tanfovx"], a["tanfovy"], H, W, shs, sh_degree, cam,
            False, False, need_backward=False, extra=(extra, None, bg_d[0:1].expand(nx)))
        out = {"rgb": rgb, "xyz_w": ex[:, 0:3], "hitmap": ex[:, 3:4].expand(num_q, 3, H, W),
               "normal": None if nv is None else ex[:, 4:7]}
    else:
        counts, rgb, radii, geom, binning, img = _native.rasterize_gaussians_batch(
            bg_d, means3D, e, opacities, sc, rotations, 1.0, e, view, proj, a["tanfovx"], a["tanfovy"], H, W, shs, sh_degree, cam,
            False, False, need_backward=False)

        def again(colors):
            return _native.recolor(bg_d, means3D, colors, e, 0, cam, H, W, counts, geom, binning, img).reshape(num_q, 3, H, W)

        out = {"rgb": rgb, "xyz_w": again(means3D), "hitmap": again(torch.ones_like(means3D)),
               "normal": None if nv is None else again(nv.contiguous())}
    return {k: (None if v is None else _finish(v, 1, num_q, h, w, super_sample_rate)) for k, v in out.items()}
