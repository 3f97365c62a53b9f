"""Instrumentation run (GSR_EXTRA_FLAGS=-DGSR_STATS): forward render at 1 and 12 views per call -- summed wave time, the
longest-lived wave and the kernel's duration (how much of the launch is the tail of the deepest list walk)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-pcloud-render_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from pcrender import camera, synth
from diff_gaussian_rasterization import _native as N
import test_gpu_batch as TB
dev = torch.device("cuda:0")
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
views = camera.circle_views(12, fov_deg=45.0, width_px=1920, height_px=1080)
out = (C.c_ulonglong * 8)()
for V in (1, 12):
    args = TB._batch_args(g, views[:V], 1920, 1080, dev)
    for _ in range(2):
        N.rasterize_gaussians_batch(*args, need_backward=True)
    torch.cuda.synchronize()
    N.lib.gsr_debug_fwd_times(out, 1)
    N.set_profiling(True)
    N.rasterize_gaussians_batch(*args, need_backward=True)
    torch.cuda.synchronize()
    prof = dict(N.get_profile()); N.set_profiling(False)
    N.lib.gsr_debug_fwd_times(out, 0)
    life, wait, stage, ev, waves, rounds, pairs, longest = [int(out[i]) for i in range(8)]
    k = prof["render_forward"]
    print("V=%d: kernel %.3f ms; %d waves, summed wave time %.1f ms (mean occupancy %.0f waves of 5120 slots (82 VGPRs: 5 waves per SIMD)), longest wave %.3f ms, mean wave %.1f us; rounds %d pairs %d"
          % (V, k, waves, life * 1e-5, life * 1e-5 / k, longest * 1e-5, life * 0.01 / waves, rounds, pairs))
