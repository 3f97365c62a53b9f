"""Instrumentation build (-DGSR_STATS): of the pairs the forward render evaluates, how many are evaluated while at most 8 / 16 / 32
of the quadrant's 64 pixels are still live?  And the same for the longest-lived waves (the launch's tail).  One view per launch
and 12 views per launch of the benchmark workload."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
out8 = (C.c_ulonglong * 8)()
NREC = 1 << 19
buf = (C.c_uint * (NREC * 10))()
N.lib.gsr_debug_fwd_records.argtypes = [C.POINTER(C.c_uint), C.c_int]
for V in (1, 12):
    args = TB._batch_args(g, views[:V], 1920, 1080, dev)
    for _ in range(2):
        N.rasterize_gaussians_batch(*args, need_backward=True)
    torch.cuda.synchronize()
    N.lib.gsr_debug_fwd_times(out8, 1)
    N.rasterize_gaussians_batch(*args, need_backward=True)
    torch.cuda.synchronize()
    n = N.lib.gsr_debug_fwd_records(buf, NREC)
    r = np.frombuffer(buf, dtype=np.uint32)[:n * 10].reshape(n, 10).astype(np.int64)
    r = r[r[:, 4] > 0]
    life, pairs, packed = r[:, 0], r[:, 6], r[:, 7]
    le8, le16, le32 = packed & 1023, (packed >> 10) & 1023, (packed >> 20) & 1023
    tot = pairs.sum()
    print("V=%d: %d waves, %d pairs evaluated; while <= 8 / 16 / 32 pixels live: %.1f %% / %.1f %% / %.1f %%"
          % (V, r.shape[0], tot, 100.0 * le8.sum() / tot, 100.0 * le16.sum() / tot, 100.0 * le32.sum() / tot))
    top = np.argsort(-life)[:64]
    print("      the 64 longest-lived waves: life %.0f-%.0f us, pairs %d-%d; of their pairs <= 8 / 16 / 32 live: %.1f %% / %.1f %% / %.1f %%"
          % (life[top].min() * 0.01, life[top].max() * 0.01, pairs[top].min(), pairs[top].max(),
             100.0 * le8[top].sum() / pairs[top].sum(), 100.0 * le16[top].sum() / pairs[top].sum(), 100.0 * le32[top].sum() / pairs[top].sum()))
