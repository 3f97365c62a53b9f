"""Instrumentation run (library built with GSR_EXTRA_FLAGS=-DGSR_STATS): where an emission workgroup's time goes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-pcloud-render_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from pcrender import camera, synth
from diff_gaussian_rasterization import _native as N
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_batch as TB
dev = torch.device("cuda:0")
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
views = camera.circle_views(12, fov_deg=45.0, width_px=1920, height_px=1080)
out = (C.c_ulonglong * 8)()
for V in (1, 12):
    args = TB._batch_args(g, views[:V], 1920, 1080, dev)
    for _ in range(2):
        N.rasterize_gaussians_batch(*args, need_backward=False)
    torch.cuda.synchronize()
    N.lib.gsr_debug_dup_times(out, 1)
    N.rasterize_gaussians_batch(*args, need_backward=False)
    torch.cuda.synchronize()
    N.lib.gsr_debug_dup_times(out, 0)
    n = max(int(out[4]), 1)
    us = [int(out[i]) * 0.01 / n for i in range(4)]
    print("V=%d: workgroups %d; mean us per workgroup: ticket %.2f, order+gather+scan %.2f, look-back %.2f, emission %.2f (sum %.2f)"
          % (V, n, us[0], us[1], us[2], us[3], sum(us)))
# cross-check against the stage's own duration
N.set_profiling(True)
args = TB._batch_args(g, views[:12], 1920, 1080, dev)
N.lib.gsr_debug_dup_times(out, 1)
N.rasterize_gaussians_batch(*args, need_backward=False)
torch.cuda.synchronize()
N.lib.gsr_debug_dup_times(out, 0)
prof = dict(N.get_profile()); N.set_profiling(False)
ticks = sum(int(out[i]) for i in range(4))
print("duplicate stage %.3f ms; summed workgroup lifetimes %d ticks over %d workgroups; if 2048 were resident throughout: %.1f ns per tick"
      % (prof["duplicate"], ticks, int(out[4]), prof["duplicate"] * 1e6 * 2048 / ticks))

# the LAST scatter launch of the forward above = second pass of the tile sort (6-bit digit)
N.lib.gsr_debug_scatter_times(out, 0)
n = max(int(out[5]), 1)
us = [int(out[i]) * 0.01 / n for i in range(5)]
print("tile-sort scatter (last pass): workgroups %d; mean us per workgroup: loads %.2f, ranking %.2f, prefix %.2f, LDS reorder %.2f, stores %.2f (sum %.2f); stage tile_sort %.3f ms"
      % (n, us[0], us[1], us[2], us[3], us[4], sum(us), prof["tile_sort"]))

# forward render waves (atomics at wave exit only: 390 K per launch, after the work)
N.lib.gsr_debug_fwd_times(out, 1)
N.set_profiling(True)
N.rasterize_gaussians_batch(*args, need_backward=True)
torch.cuda.synchronize()
prof = dict(N.get_profile()); N.set_profiling(False)
N.lib.gsr_debug_fwd_times(out, 0)
life, wait, stage, ev, waves, rounds, pairs = [int(out[i]) for i in range(7)]
print("forward render: %d waves, %d rounds, %d pairs; wave time %.1f ms-waves: waiting for records %.1f %%, footprint test + staging %.1f %%, pair evaluation %.1f %%, rest %.1f %%; "
      "%.2f us per round waiting, %.3f us per pair; stage %.3f ms"
      % (waves, rounds, pairs, life * 1e-5, 100.0 * wait / life, 100.0 * stage / life, 100.0 * ev / life, 100.0 * (life - wait - stage - ev) / life,
         wait * 0.01 / max(rounds, 1), ev * 0.01 / max(pairs, 1), prof["render_forward"]))

# backward render waves
counts, color, radii, geom, binning, img = N.rasterize_gaussians_batch(*args, need_backward=True)
dL = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, (12, 3, 1080, 1920)).astype(np.float32)).to(dev)
N.lib.gsr_debug_bwd_times(out, 1)
N.set_profiling(True)
N.rasterize_gaussians_backward_batch(args[0], args[1], radii, args[2], args[4], args[5], 1.0, args[7], args[8], args[9], args[10], args[11], dL,
                                     args[14], args[15], args[16], geom, binning, img, False)
torch.cuda.synchronize()
prof = dict(N.get_profile()); N.set_profiling(False)
N.lib.gsr_debug_bwd_times(out, 0)
life, wait, setup, stage, ev, longest, groups, packed = [int(out[i]) for i in range(8)]
items, waves = packed >> 20, packed & ((1 << 20) - 1)
k = prof["render_backward"]
print("backward render: %d items over %d waves, %d groups of 4 entries; wave time %.1f ms-waves: item set-up %.1f %%, footprint test + staging %.1f %%, groups %.1f %%, "
      "wait at rotation %.1f %%, rest %.1f %%; %.2f us per group; stage %.3f ms, mean wave life %.3f ms, longest %.3f ms, mean occupancy %.0f waves"
      % (items, waves, groups, life * 1e-5, 100.0 * setup / life, 100.0 * stage / life, 100.0 * ev / life, 100.0 * wait / life,
         100.0 * (life - wait - setup - stage - ev) / life, ev * 0.01 / max(groups, 1), k, life * 1e-5 / max(waves, 1), longest * 1e-5, life * 1e-5 / k))
