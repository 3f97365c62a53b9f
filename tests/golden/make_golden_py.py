"""Generate golden vectors from the reference's own PYTHON code (run in the build container, where
/root/reference exists; the GPU box never runs this).

  py_sh_eval.npz : /root/reference/models/sh_utils.py::eval_sh (pure torch) on seeded inputs, degrees 0..3,
                   plus RGB2SH.  Pins the SH basis / constants of the rasterizer's colour path.
  py_camera.npz  : the `pcrender --cam_mode circle` camera (simple_benchmark.py:146-149) through the reference's
                   generate_cam -> CameraTrajectory -> Camera and get_rasterize_param_from_camera
                   (simple_raw_render.py:17-49,79-112), and the reference's only shipped fixture
                   validate/temp_state_dict.pt.  Third-party modules the reference imports but this path never
                   calls (MinkowskiEngine, open3d, cv2, ...) are replaced by MagicMock at import time.

  py_uvmap.npz   : /root/reference/plib/uv_mapping.py::UVMap (the texture lookup of the mesh sampler, through scipy's
                   RegularGridInterpolator) on a seeded 7x5 RGB texture: uv inside, on the texel centres, on the borders, and
                   outside [0,1) (wrap).  Pins pcrender.mesh_sample.uv_lookup.

  py_rasterize_calls.npz : the CALL TRACE of the reference's caller glue.  PCML_Render.render (simple_raw_render.py:290-524: the
                   four passes world-xyz / SH colour / hit map / normals, each through PCML_Render._rasterize, :227-288) and
                   Simple_Render.render (:662-854: three passes) are driven as they stand on a seeded toy input (24 points, four
                   circle views of 8x6 px, super-sample 2) with a RECORDING stand-in for GaussianRasterizer that returns seeded fake
                   images: what is stored is every argument of every rasterizer call (scales after the sqrt(3)/scale_factor*6
                   factor, colors_precomp after the per-view normal-sign flip of :264-268, the settings built per call) and the
                   dictionaries render() returns from the fake images (stack, bilinear down-filter, permute).  The network, its
                   checkpoint and MinkowskiEngine are not on this path's arithmetic: the model is a stub that returns the seeded
                   primitives, PCML_Render.__init__ (checkpoint loading) is bypassed.  Pins pcrender.raster_passes.

Usage: python tests/golden/make_golden_py.py [sh|camera|uvmap|calls ...]
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def sh_vectors():
    sys.path.insert(0, REF)
    from models import sh_utils
    rng = np.random.default_rng(2024)
    out = {}
    for deg in range(4):
        K = (deg + 1) ** 2
        n = 64
        sh = rng.standard_normal((n, 3, K)).astype(np.float32)          # eval_sh wants [..., C, K]
        d = rng.standard_normal((n, 3))
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        res = sh_utils.eval_sh(deg, torch.from_numpy(sh), torch.from_numpy(d)).numpy()
        out["deg%d_sh" % deg] = sh
        out["deg%d_dirs" % deg] = d
        out["deg%d_result" % deg] = res.astype(np.float32)
    rgb = rng.uniform(0, 1, (32, 3)).astype(np.float32)
    out["rgb"] = rgb
    out["rgb2sh"] = sh_utils.RGB2SH(torch.from_numpy(rgb)).numpy()
    np.savez_compressed(os.path.join(OUT, "py_sh_eval.npz"), **out)
    print("py_sh_eval.npz written")


def _import_reference_caller():
    """simple_raw_render with the third-party modules it never calls on this path replaced by mocks"""
    if "simple_raw_render" in sys.modules:
        return sys.modules["simple_raw_render"]
    for name in ["MinkowskiEngine", "open3d", "imageio", "cv2", "torch_scatter", "xatlas", "skimage", "skimage.metrics",
                 "pyexr", "matplotlib", "matplotlib.pyplot", "matplotlib.font_manager", "mpl_toolkits",
                 "mpl_toolkits.axes_grid1", "lpips", "pytorch_msssim", "tqdm", "diff_gaussian_rasterization"]:
        sys.modules[name] = MagicMock()
    from typing import NamedTuple

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

    sys.modules["diff_gaussian_rasterization"].GaussianRasterizationSettings = GaussianRasterizationSettings
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import simple_raw_render as srr
    finally:
        os.chdir(cwd)
    return srr


def camera_vectors():
    srr = _import_reference_caller()
    out = {}
    fix = torch.load(os.path.join(REF, "validate", "temp_state_dict.pt"))
    out["fixture_H_c2w"] = fix["H_c2w"].numpy()
    out["fixture_intrinsic"] = fix["intrinsic"].numpy()
    for tag, (w, h, fov, ss) in {"native": (512, 512, 45.0, 2), "hd": (1920, 1080, 45.0, 1), "fov60": (640, 360, 60.0, 1)}.items():
        cam_info = {'fov': fov, 'width_px': w, 'height_px': h, 'mode': 'circle', 'n_imgs': 12, 'd': 0, 'r': 3,
                    'center_angles': [90, 0], 'alt_yaxis': False}
        camera = srr.generate_cam(cam_info, save_temp_state_dict=False)
        out[tag + "_H_c2w"] = camera.H_c2w.numpy()
        chunks = camera.chunk(12, dim=1) if hasattr(camera, "chunk") else None
        views, projs, campos, tans, sizes = [], [], [], [], []
        for j in range(12):
            s = srr.get_rasterize_param_from_camera(chunks[j], device=torch.device("cpu"), fovX_deg=fov, fovY_deg=fov,
                                                    sh_degree=1, bg=None, super_sample_rate=ss)
            views.append(s.viewmatrix.contiguous().numpy().reshape(4, 4))
            projs.append(s.projmatrix.contiguous().numpy().reshape(4, 4))
            campos.append(s.campos.numpy().reshape(3))
            tans.append([s.tanfovx, s.tanfovy])
            sizes.append([s.image_height, s.image_width])
        out[tag + "_viewmatrix"] = np.stack(views)
        out[tag + "_projmatrix"] = np.stack(projs)
        out[tag + "_campos"] = np.stack(campos)
        out[tag + "_tanfov"] = np.array(tans, dtype=np.float64)
        out[tag + "_size"] = np.array(sizes, dtype=np.int64)
        out[tag + "_args"] = np.array([w, h, fov, ss], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "py_camera.npz"), **out)
    print("py_camera.npz written")


def uvmap_vectors():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_uv_mapping", os.path.join(REF, "plib", "uv_mapping.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(77)
    tex = rng.uniform(0, 1, (7, 5, 3)).astype(np.float32)
    uv = np.concatenate([
        rng.uniform(0, 1, (200, 2)),                                       # inside
        rng.uniform(-2.5, 3.5, (100, 2)),                                  # outside: wrapped
        (np.stack(np.meshgrid(np.arange(5), np.arange(7)), -1).reshape(-1, 2) + 0.5) / np.array([5.0, 7.0]),   # texel centres
        np.array([[0.0, 0.0], [1.0, 1.0], [0.0, 0.999999], [0.999999, 0.0], [0.1, 0.0], [0.0, 0.07], [0.5, 1.0 - 1e-9]]),
    ], 0)
    out = mod.UVMap(tex)(uv)
    np.savez_compressed(os.path.join(OUT, "py_uvmap.npz"), texture=tex, uv=uv, result=np.asarray(out, np.float64))
    print("py_uvmap.npz written", out.shape)


def call_trace_vectors():
    srr = _import_reference_caller()
    rng = np.random.default_rng(4242)
    n, h, w, ss, fov, sf, offset = 24, 6, 8, 2, 45.0, 256.0, 512
    calls = []

    class RecordingRasterizer:
        """stands where diff_gaussian_rasterization.GaussianRasterizer stands in the reference's module"""
        def __init__(self, raster_settings):
            self.st = raster_settings

        def __call__(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
            k = len(calls)
            st = self.st
            rec = dict(means3D=means3D, means2D=means2D, opacities=opacities, shs=shs, colors_precomp=colors_precomp, scales=scales,
                       rotations=rotations, cov3D_precomp=cov3D_precomp, viewmatrix=st.viewmatrix.contiguous().reshape(4, 4),
                       projmatrix=st.projmatrix.contiguous().reshape(4, 4), campos=st.campos.reshape(3), bg=st.bg,
                       scalars=torch.tensor([st.image_height, st.image_width, st.tanfovx, st.tanfovy, st.scale_modifier, st.sh_degree,
                                             float(st.prefiltered), float(st.debug)], dtype=torch.float64))
            calls.append({a: (None if v is None else v.detach().clone().numpy()) for a, v in rec.items()})
            img = torch.from_numpy(np.random.default_rng(9000 + k).random((3, st.image_height, st.image_width), dtype=np.float32))
            return img, torch.zeros(means3D.shape[0], dtype=torch.int32)

    srr.GaussianRasterizer = RecordingRasterizer
    # the caller writes device="cuda" literally (simple_raw_render.py:241) and synchronises around its timers: run it on the CPU
    zeros_like = torch.zeros_like
    torch.zeros_like = lambda *a, **k: zeros_like(*a, **{**k, "device": "cpu"} if k.get("device") == "cuda" else k)
    torch.cuda.synchronize = lambda *a, **k: None
    cam = srr.generate_cam({'fov': fov, 'width_px': w, 'height_px': h, 'mode': 'circle', 'n_imgs': 4, 'd': 0, 'r': 3,
                            'center_angles': [90, 0], 'alt_yaxis': False}, save_temp_state_dict=False)
    out = {"H_c2w": cam.H_c2w.numpy(), "args": np.array([n, h, w, ss, fov, sf, offset], dtype=np.float64)}

    # ---- PCML_Render.render: the network is a stub handing back seeded primitives (voxel coordinates, like the decoder's output)
    vox = torch.from_numpy(rng.integers(400, 624, (n, 3)).astype(np.float32))
    prim = dict(
        decoded_primitives=[vox], decoded_sh=[torch.from_numpy(rng.standard_normal((n, 13, 3)).astype(np.float32))],
        decoded_r=[torch.from_numpy((np.array([1, 0, 0, 0]) + 0.05 * rng.standard_normal((n, 4))).astype(np.float32))],
        decoded_s=[torch.from_numpy(np.clip(1 + 0.15 * rng.standard_normal((n, 3)), 0, None).astype(np.float32))],
        decoded_o=[torch.from_numpy(rng.uniform(0.2, 1.0, (n, 1)).astype(np.float32))],
        decoded_n=[torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32))])
    for k, v in prim.items():
        out["pcml_" + k] = v[0].numpy()
    model_out = (prim["decoded_primitives"], prim["decoded_sh"], prim["decoded_r"], prim["decoded_s"], prim["decoded_o"], 0.0,
                 None, None, None, 0.0, 0.0, prim["decoded_n"])
    r = object.__new__(srr.PCML_Render)            # __init__ loads a checkpoint: not on this path
    r.device = torch.device("cpu")
    r.model = lambda sparse: model_out
    r.info = {"clr_encoder_channels": "3 64", "sh_deg": 1, "scale_factor": sf}
    r.voxelized, r.scale_factor, r.offset = True, sf, offset
    srr.ME.utils.sparse_collate = lambda pts, feats: (pts, feats)

    class _Pcd:
        xyz_w = [vox.clone()]
        rgb = [torch.from_numpy(rng.uniform(0, 1, (n, 3)).astype(np.float32))]

    ret = r.render(_Pcd(), 1, cam, fov, enable_opacity=True, super_sample_rate=ss, background_color=1.0)
    n_pcml = len(calls)
    for k in ("xyz_w", "rgb", "hitmap", "normal"):
        out["pcml_ret_" + k] = ret[k].numpy()

    # ---- Simple_Render.render (no network): isotropic splats from positions + colours
    sr = srr.Simple_Render(voxelized=True, scale_factor=sf, offset=offset)
    sr.device = torch.device("cpu")
    sr.default_quaternion = sr.default_quaternion.cpu()
    out["simple_xyz"], out["simple_rgb"] = _Pcd.xyz_w[0].numpy(), _Pcd.rgb[0].numpy()
    ret2 = sr.render(_Pcd(), 1, cam, fov, enable_opacity=True, super_sample_rate=ss, background_color=0.0, sigma=1.5)
    for k in ("rgb", "xyz_w", "hitmap"):
        out["simple_ret_" + k] = ret2[k].numpy()
    assert ret2["normal"] is None
    out["n_calls"] = np.array([n_pcml, len(calls) - n_pcml])
    for k, c in enumerate(calls):
        for a, v in c.items():
            if v is not None:
                out["call%02d_%s" % (k, a)] = v
    torch.zeros_like = zeros_like
    np.savez_compressed(os.path.join(OUT, "py_rasterize_calls.npz"), **out)
    print("py_rasterize_calls.npz written: %d + %d calls" % (n_pcml, len(calls) - n_pcml))


if __name__ == "__main__":
    todo = sys.argv[1:] or ["sh", "camera", "uvmap", "calls"]
    if "sh" in todo:
        sh_vectors()
    if "camera" in todo:
        camera_vectors()
    if "uvmap" in todo:
        uvmap_vectors()
    if "calls" in todo:
        call_trace_vectors()
