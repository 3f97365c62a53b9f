"""Instrumentation run (GSR_EXTRA_FLAGS=-DGSR_STATS): where the waves of the cooperative forward kernel (one view per call) spend
their time, per role.  usage: python scripts/coop_times.py"""
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
out = (C.c_ulonglong * 32)()
for v in (0, 1, 5):
    args = TB._batch_args(g, views[v:v + 1], 1920, 1080, dev)
    for _ in range(2):
        N.rasterize_gaussians_batch(*args, need_backward=True)
    torch.cuda.synchronize()
    N.lib.gsr_debug_coop_times(out, 1)
    N.set_profiling(True)
    N.rasterize_gaussians_batch(*args, need_backward=True)
    torch.cuda.synchronize()
    prof = dict(N.get_profile()); N.set_profiling(False)
    N.lib.gsr_debug_coop_times(out, 0)
    print("view %d: kernel %.3f ms" % (v, prof["render_forward"]))
    for r, name, labels in ((0, "consumer", ("life", "barrier", "blend", "-", "-", "iterations", "pairs")),
                            (1, "stager", ("life", "barrier", "form", "gather wait", "eval", "iterations", "rounds")),
                            (2, "producer", ("life", "barrier", "eval", "-", "-", "iterations", "pairs"))):
        o = [int(out[8 * r + i]) for i in range(8)]
        if o[7] == 0:
            continue
        print("  %-9s %6d waves, mean life %7.2f us, longest %7.2f us | " % (name, o[7], o[0] * 0.01 / o[7], int(out[24 + r]) * 0.01) +
              ", ".join("%s %.1f%%" % (labels[i], 100.0 * o[i] / max(o[0], 1)) for i in (1, 2, 3, 4) if labels[i] != "-") +
              " | %s %.1f, %s %.1f per wave" % (labels[5], o[5] / o[7], labels[6], o[6] / o[7]))
    print("  consumer timeline: last end %.1f us after the first start; mean start %.1f us, last start %.1f us; %d waves live > 80 us, their mean start %.1f us"
          % (int(out[27]) * 0.01, int(out[30]) * 0.01, int(out[31]) * 0.01, int(out[29]), int(out[28]) * 0.01))
