"""Instrumentation run (library built with GSR_EXTRA_FLAGS="-DGSR_STATS -DGSR_STATS_HITS"): how much of the render backward's staged work hits."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-pcloud-render_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import util
from pcrender import camera, synth
from diff_gaussian_rasterization import _native as N
dev = torch.device("cuda:0")
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
gg = synth.make_gaussians(cloud, profile="training", seed=1)
out = (C.c_ulonglong * 8)()
for vid in (0, 3):
    v = camera.circle_views(12, fov_deg=45.0, width_px=1920, height_px=1080)[vid]
    s = util.scene_from(gg, v, 1920, 1080, bg=(1, 1, 1))
    N.lib.gsr_debug_bwd_stats(out, 1)
    util.run_product(s, dev, dL_dpix=util.seeded_dL(s), light=True)
    torch.cuda.synchronize()
    N.lib.gsr_debug_bwd_stats(out, 0)
    r, st, gr, grh, eh, ph = [int(x) for x in out[:6]]
    print("view %d: rounds %d staged entries %d (%.1f / round) groups %d with-hit %d (%.0f%%) entries-with-hit %d (%.0f%% of staged) pixel-hits %d (%.1f per hit entry)"
          % (vid, r, st, st / max(r, 1), gr, grh, 100.0 * grh / max(gr, 1), eh, 100.0 * eh / max(st, 1), ph, ph / max(eh, 1)))
    print("        batches flushed %d (%.2f groups per batch), 2x2 pixel blocks hit by some entry of the batch: %.2f of 16"
          % (int(out[6]), grh / max(int(out[6]), 1), int(out[7]) / max(int(out[6]), 1)))
