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

Usage: python tests/golden/make_golden_py.py [sh|camera|uvmap ...]
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


def camera_vectors():
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


if __name__ == "__main__":
    todo = sys.argv[1:] or ["sh", "camera", "uvmap"]
    if "sh" in todo:
        sh_vectors()
    if "camera" in todo:
        camera_vectors()
    if "uvmap" in todo:
        uvmap_vectors()
