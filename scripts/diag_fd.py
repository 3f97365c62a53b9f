"""Diagnostic: step-size dependence of the single-Gaussian finite-difference slope (tests/test_gpu_configs.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-pcloud-render_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import util
import test_gpu_configs as T
from pcrender import camera, synth
dev = torch.device("cuda:0")
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
gg = synth.make_gaussians(cloud, profile="training", seed=1)
v = camera.circle_views(12, fov_deg=45.0, width_px=1920, height_px=1080)[3]
s = util.scene_from(gg, v, 1920, 1080, bg=(1, 1, 1))
dL = util.seeded_dL(s)
_, g = util.run_product(s, dev, dL_dpix=dL, light=True)
for field, key in (("means3D", "dL_dmean3D"), ("scales", "dL_dscale")):
    for eps in (0.2, 0.05, 0.0125, 0.003):
        rng = np.random.default_rng(77)
        T._fd_single_gaussians(s, dev, g, dL, field, key, eps, 32, rng, "eps=%g" % eps, in_plane=True)
