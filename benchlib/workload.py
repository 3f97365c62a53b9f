"""What the bench renders and what it has to move: BASELINE.json's configs, the view a rank renders at a step, the
algorithmic-bytes model of SURVEY.md 8(d), the identity of the kernel sources."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels_sha():
    """identity of the kernel sources this process runs (csrc/*.hip, *.hpp + include/gsr.h): counter profiles taken on other
    kernels are not quoted (roofline.traffic is null then)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gaussian-pcloud-render_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../include/gsr.h"]:
        if f.endswith((".hip", ".hpp", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]



def algorithmic_bytes(P, V, R, T, N, K, C_fwd, C_bwd, tile_passes, views_per_call=1):
    """Minimum HBM bytes each stage has to move PER FRAME when `views_per_call` views of one cloud travel in one submission.
    Render / preprocess figures are SURVEY.md 8(d)'s per-view formulas with what a batched launch shares counted once per batch:
    the per-Gaussian inputs of preprocess ((44 + 12 K) P) and of the per-Gaussian backward ((107 + 12 K) V) are read once for all
    views, and the backward's outputs ((40 + 12 K) V) are written once (summed over the views in registers).  The sort figures
    follow this library's own data flow (u32 depth keys and ids; tile keys are u16 for images of up to 65 536 tiles); the tile
    ranges are found by binary search (about 24 two-byte probes per tile), not by a pass over the keys, and the prefix sum of the
    tile counts is part of the pair emission (no separate scan).  R here is the number of pairs IN THE LISTS (`list_pairs_avg` of
    the line: the reference's num_rendered minus the pairs footprint clipping leaves out, include/gsr.h reference_lists); C_fwd /
    C_bwd are the list entries the render kernels consume, counted in those lists."""
    b = {}
    kb = 2 if T <= 65536 else 4
    vpc = max(1, int(views_per_call))
    b["preprocess"] = (44 + 12 * K) * P / vpc + 75 * V + 8 * (P - V)
    b["depth_sort"] = 4 * (4 + 8 + 8) * P
    b["duplicate"] = 20 * P + (kb + 4) * R
    b["tile_sort"] = tile_passes * (kb + 2 * (kb + 4)) * R
    b["tile_ranges"] = (24 * kb + 16) * T
    b["render_forward"] = 40 * C_fwd + 8 * T + 20 * N
    b["render_backward"] = 40 * C_bwd + 20 * N + 44 * V
    b["preprocess_backward"] = 92 * V + ((107 + 12 * K) * V + (40 + 12 * K) * V) / vpc
    return b



# BASELINE.json configs this bench can run (`--config`); 2 is the headline
CONFIGS = {
    1: dict(workload="synth-THuman-256", width=1920, height=1080, forward_only=True, n_views=12, shard="circle",
            what="configs[1]: THuman-256 (200K voxelised points), 1080p, forward-only"),
    2: dict(workload="synth-THuman-800K", width=1920, height=1080, forward_only=False, n_views=12, shard="circle",
            what="configs[2]: THuman-800K, 1080p, forward+backward (the headline)"),
    3: dict(workload="synth-THuman-800K", width=1920, height=1080, forward_only=False, n_views=8, shard="views",
            what="configs[3]: THuman-800K, 8 camera views sharded across the ranks, frames gathered on rank 0"),
    4: dict(workload="synth-mesh-2M", width=3840, height=2160, forward_only=False, n_views=8, shard="views",
            what="configs[4]: 2M-point sampled mesh, 3840x2160, forward+backward, 8 camera views sharded across the ranks"),
}



def view_of(k, rank, world, n_views, shard):
    """Camera view of local step k on `rank`.
    shard "circle" (weak scaling of the headline): every rank walks the whole circle, rank r starting (r n_views) // world views
    in, so ANY n_views consecutive steps of a rank are n_views distinct views whatever the world size, and at a given step the
    ranks render different views (world <= n_views).
    shard "views" (configs[3] / [4]): the n_views views of one turn are dealt round-robin, rank r owns {v : v mod world == r}
    (pcrender.multiview.shard_views) and walks its own views in order."""
    if shard == "circle":
        return (k + (rank * n_views) // world) % n_views
    mine = list(range(rank, n_views, world)) or [rank % n_views]
    return mine[k % len(mine)]


