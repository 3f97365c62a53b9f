"""bench.py -- frames/s of the rasterizer hot path (forward + backward) on the headline workload.

    python bench.py [--gpus N --steps K --warmup W]        (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json metric, configs[2]): synth-THuman-800K -- 800 000 synthetic Gaussians
("training" profile: opacity U(0.2,1), SH degree 1 in 13 rows) rendered at 1920x1080 from the reference's 12
`circle` cameras, forward + backward through the public API, loss = sum(img * G).
A step = one frame (one camera view) per rank; views are sharded round-robin over ranks, frames are gathered on rank 0 with
RCCL, one collective per submission, on a stream of its own (weak scaling).  All inputs are resident in HBM before the timed region.

HEADLINE call shape (`config.workload` says so): frames are submitted --views-per-call (default 12, one turn of the circle) at a
time through rasterize_views -- the C ABI's gsr_forward_batch / gsr_backward_batch: every kernel covers all views of the
submission, nothing on the host waits for the device inside a frame, gradients of the shared cloud are summed over the views on
the device -- with --streams (default 1) submissions in flight on their own HIP streams.  K steps that are not a multiple of the
batch end with a smaller batch.  The timed region is `--repeats` (default 5) blocks of exactly K steps, every block bracketed by
barrier + synchronize; `value` / `ms_per_step` are the MEDIAN block (max over ranks per block), all blocks are listed.

Beside the headline the line carries (rank 0, measured right after the timed region, never inside it):
  kernels_ms / roofline -- a single-stream pass with hipEvents around every stage on the launch stream: the dominant kernel's
                  ALGORITHMIC bytes / its measured duration against the 8 TB/s HBM peak (and the ~6.3 TB/s a copy reaches), its
                  VALU issue rate, and -- only if profiles/pmc_traffic.json was taken on exactly this workload and call shape -- the
                  counter traffic and the profile's own duration of that kernel (`profile_avg_ms`, `frac_profile`,
                  `live_vs_profile`); `traffic` is null otherwise;
  drop_in      -- the CO-HEADLINE: the same frames through the reference's API exactly as its caller uses it
                  (/root/reference/simple_raw_render.py:259-278): per view, fresh settings tensors built inside the timed loop, one
                  GaussianRasterizer(settings)(...) call, loss.backward(); one thread, one stream, in order; plus the same figure
                  from --drop-in-processes (default 5) FRESH processes (min / median / max);
  drop_in_api  -- variants of it (prebuilt settings; the opt-in overlap of consecutive calls; four caller threads) and the
                  per-stage times of the per-view call;
  forward_only -- inference frames/s of both call shapes;
  rgb_time_equiv -- the reference's own timing hook (`rgb time`, simple_raw_render.py:433-456): 12 views x 1024^2 (512^2 camera,
                  super-sample 2), the SH colour pass, INCLUDING per-view settings glue and the bilinear down-filter;
  cpu_baseline -- the plain-C oracle (OpenMP) on frames of the same workload (N=1): 1 warm-up + median of 5 frames on all host
                  cores, plus a 1-core figure from one frame.

`--gpus N` with N > 1 and no torch.distributed.run environment re-launches itself under torch.distributed.run with N ranks
(one per GPU, RCCL); it refuses loudly when the box has fewer than N GPUs.
"""
import argparse
import json
import os
import sys
import time

# Each frame in flight needs a hardware queue of its own: the ROCm runtime maps HIP streams onto GPU_MAX_HW_QUEUES
# (default 4) hardware queues, and with four worker streams plus the stream the frames are gathered on, two of them end
# up sharing one queue and serialise (measured: 800 vs 900 frames/s on the same box).  Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# RCCL / cross-process device memory on this pool needs dmabuf IPC (the image exports it already; kept for any other launcher)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "gaussian-pcloud-render_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from benchlib import cpu_baseline as _cpu_baseline, dist as _dist, roofline as _roofline  # noqa: E402
from benchlib.roofline import HBM_ACHIEVABLE_GBS, HBM_PEAK_GBS, STAGE_KERNEL, VALU_PEAK_GWIPS, stage_kernel  # noqa: E402,F401
from benchlib.timing import GC_RECOVER_S, union_ms  # noqa: E402,F401
from benchlib.workload import CONFIGS, algorithmic_bytes, kernels_sha, view_of  # noqa: E402,F401

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="BASELINE.json configs[i] (2 = the headline): sets the workload, image size, forward-only and how views are "
                         "dealt to the ranks; --workload / --width / --height / --forward-only still override")
    ap.add_argument("--workload", default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--warmup-seconds", type=float, default=1.0,
                    help="untimed frames are rendered for at least this long on top of --warmup, so that the timed region starts at "
                         "the clocks the chip sustains (disclosed in warmup_effective)")
    ap.add_argument("--gather-mode", default="collective", choices=["collective", "p2p"],
                    help="frame gather to rank 0: one dist.gather per submission, or a grouped irecv / isend per peer "
                         "(dist.batch_isend_irecv; DESIGN.md section 8)")
    ap.add_argument("--points", type=int, default=None, help="override the point count (debug)")
    ap.add_argument("--profile", default="training", choices=["training", "inference"])
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--views-per-call", type=int, default=int(os.environ.get("GSR_BENCH_VIEWS_PER_CALL", "0")),
                    help="frames submitted per rasterizer call: 1 = the reference's per-view GaussianRasterizer call; V > 1 = "
                         "rasterize_views (C ABI gsr_forward_batch / gsr_backward_batch), V views of the cloud in one submission; "
                         "0 = the config's default: 12 (one turn of the circle) for configs 1 / 2, the rank's own views for 3 / 4")
    ap.add_argument("--no-per-view", action="store_true", help="skip the drop-in-API / forward-only / rgb-time side measurements")
    ap.add_argument("--drop-in-processes", type=int, default=5,
                    help="fresh processes that each measure the drop-in figure on their own (min / median / max in the line: the "
                         "figure must not depend on how a process happened to set up its streams); 0 = skip")
    ap.add_argument("--drop-in-probe", action="store_true", help=argparse.SUPPRESS)   # what those processes run
    ap.add_argument("--gather-probe", action="store_true", help=argparse.SUPPRESS)    # 1-rank RCCL anchor, run as a child under RANK=0 WORLD_SIZE=1
    ap.add_argument("--no-stage-events", action="store_true",
                    help="no per-stage hipEvents anywhere (for timeline traces: an event pair costs ~10 us of bubble per stage)")
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps each; the median block is reported")
    ap.add_argument("--cpu-frames", type=int, default=5, help="frames of the all-core CPU baseline (after 1 warm-up)")
    ap.add_argument("--no-cpu-1core", action="store_true", help="skip the 1-core CPU figure (about a minute of CPU time)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("GSR_BENCH_STREAMS", "0")),
                    help="host threads per rank, each rendering whole submissions on its own HIP stream (views are independent); "
                         "0 = 1: one submission in flight (a second one is worth +-2 %%, box noise -- the batched kernels have no tail to "
                         "hide -- and makes the in-region hipEvent duration of a kernel include whatever the other stream interleaves: "
                         "2.39-2.81 ms for a kernel whose own trace says 2.29)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --device-index 0 lets several ranks share one GPU to exercise the multi-rank control flow "
                         "(frames then travel through host memory; not a performance mode)")
    ap.add_argument("--device-index", type=int, default=-1, help="GPU of this rank (default: LOCAL_RANK)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    args.workload = args.workload or cfg["workload"]
    args.width = args.width or cfg["width"]
    args.height = args.height or cfg["height"]
    args.forward_only = args.forward_only or cfg["forward_only"]

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path in the product")
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ      # under torch.distributed.run
    if args.gpus > 1 and not launched:
        raise SystemExit(_dist.self_launch(args, os.path.abspath(__file__), torch.cuda.device_count()))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev_index = local_rank if args.device_index < 0 else args.device_index
    if dev_index >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d (local rank %d) wants GPU %d but only %d are visible: one device per rank"
                         % (rank, local_rank, dev_index, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    host_collectives = args.dist_backend == "gloo"
    use_dist = launched            # a 1-rank launch under torch.distributed.run still goes through RCCL (gather, barrier, max)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if host_collectives:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _native, rasterize_views
    from pcrender import camera, multiview, synth

    W, H = args.width, args.height
    cloud = synth.make_cloud(args.workload, seed=0, P=args.points)
    g = synth.make_gaussians(cloud, profile=args.profile, seed=1)
    P, M, D = g["means3D"].shape[0], g["shs"].shape[1], g["sh_degree"]
    n_views, shard = cfg["n_views"], cfg["shard"]
    if args.views_per_call <= 0:
        args.views_per_call = n_views if shard == "circle" else max(1, len(range(rank, n_views, world)))
    if args.streams <= 0:
        args.streams = 1
    # the reference's circle cameras (simple_raw_render.py:259-278 loops over them); configs[3] / [4] use 8 of them
    views = camera.circle_views(n_imgs=n_views, fov_deg=45.0, width_px=W, height_px=H)
    bg = torch.ones(3, device=dev)  # simple_benchmark.py:332 background (1,1,1)
    settings = [GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=v["tanfovx"], tanfovy=v["tanfovy"], bg=bg, scale_modifier=1.0,
        viewmatrix=v["viewmatrix"].to(dev), projmatrix=v["projmatrix"].to(dev), sh_degree=D, campos=v["campos"].to(dev),
        prefiltered=False, debug=False) for v in views]
    rasterizers = [GaussianRasterizer(s) for s in settings]
    from pcrender import raster_passes as _rp
    H_c2w_views = camera.circle_path(n_views, 0, 3, [90, 0])     # the poses `views` was built from (camera.circle_views)
    if world > 1:
        # one process per GPU on a shared host: keep every rank's host-side thread pools (torch CPU ops, OpenMP) to its share
        torch.set_num_threads(max(1, (os.cpu_count() or 1) // world))

    grad = not args.forward_only

    def make_leaves(with_grad):
        leaf = lambda a: torch.from_numpy(a).to(dev).requires_grad_(with_grad)  # noqa: E731
        m3 = leaf(g["means3D"])
        return dict(means3D=m3, means2D=torch.zeros_like(m3, requires_grad=with_grad), shs=leaf(g["shs"]),
                    opacities=leaf(g["opacities"]), scales=leaf(g["scales"]), rotations=leaf(g["rotations"]))

    # one set of leaf tensors per host thread (their .grad is written by that thread's backward only)
    leafsets = [make_leaves(grad) for _ in range(max(4, args.streams))]
    means3D, shs, opac = leafsets[0]["means3D"], leafsets[0]["shs"], leafsets[0]["opacities"]
    scales, rots = leafsets[0]["scales"], leafsets[0]["rotations"]
    G = torch.from_numpy(np.random.default_rng(123).uniform(-1, 1, (3, H, W)).astype(np.float32)).to(dev)
    VPC = max(1, args.views_per_call)
    # the frames of one submission travel in ONE collective (up to VPC x 24.9 MB per rank at 1080p): fewer, larger transfers
    # for the point-to-point xGMI links than a gather per frame
    gather_bufs = [torch.empty((VPC, 3, H, W), device="cpu" if host_collectives else dev) for _ in range(world)] \
        if (use_dist and rank == 0) else None
    do_gather = use_dist and not args.no_gather
    # the gather has a stream of its own: it only waits for the submission it ships, the next submission's kernels run beside it
    gather_stream = torch.cuda.Stream(device=dev) if do_gather else None
    gather_events = []      # (start, end) on the gather stream, one pair per submission of the timed region

    # ---- instrumentation INSIDE the timed region: two hipEvents around every submission on the stream it runs on (GPU time of the
    # timed steps themselves, not of a pass taken afterwards) and one shader-clock probe per submission on a side stream (the
    # clock the chip sustains under exactly this load).  Costs two event records and one 64-lane kernel per submission.
    sub_events = []          # (tag, start event, end event)
    sub_tag = [None]         # None: not recording
    probe = _native.ClockProbe(dev, capacity=1024)

    def bracket(fn):
        if sub_tag[0] is None:
            return fn()
        cur = torch.cuda.current_stream(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        probe.launch()
        out = fn()
        e1.record(cur)
        sub_events.append((sub_tag[0], e0, e1))
        return out

    def render(i, tslot=0, with_grad=True, vpc_views=None):
        """Forward (+ backward) of local step i on the calling thread's current stream; returns the frame."""
        v = view_of(i, rank, world, n_views, shard)
        L = leafsets[tslot]
        if with_grad:
            img, _ = rasterizers[v](**L)
            (img * G).sum().backward()
            for t in L.values():
                t.grad = None
            return img.detach()
        with torch.no_grad():
            img, _ = rasterizers[v](**L)
        return img

    literal_mode = [False]
    host_trace = None     # ([settings built], [C forward entered]) perf_counter stamps while drop_in_host_exposed runs

    def render_literal(i, tslot=0, with_grad=True):
        """The reference caller's loop body, literally (simple_raw_render.py:260-278): the settings of the view are BUILT for this
        call (get_rasterize_param_from_camera's arithmetic on the host, fresh device tensors), a GaussianRasterizer is made from
        them, called, and the loss is backpropagated."""
        v = view_of(i, rank, world, n_views, shard)
        L = leafsets[tslot]
        st = _rp.settings_for_view(H_c2w_views[v], W, H, 45.0, dev, sh_degree=D, bg=bg, super_sample_rate=1)
        if host_trace is not None:
            host_trace[0].append(time.perf_counter())
        if with_grad:
            img, _ = GaussianRasterizer(st)(**L)
            (img * G).sum().backward()
            for t in L.values():
                t.grad = None
            return img.detach()
        with torch.no_grad():
            img, _ = GaussianRasterizer(st)(**L)
        return img

    def render_many(i, n, tslot=0, with_grad=True):
        if literal_mode[0] and n == 1:
            return bracket(lambda: render_literal(i, tslot, with_grad)[None])
        return bracket(lambda: render_many_(i, n, tslot, with_grad))

    def render_many_(i, n, tslot=0, with_grad=True):
        """Local steps i .. i+n-1 of this rank in ONE rasterizer call (n <= --views-per-call); returns the frames [n,3,H,W]."""
        if n == 1:
            return render(i, tslot, with_grad)[None]
        L = leafsets[tslot]
        sts = [settings[view_of(i + k, rank, world, n_views, shard)] for k in range(n)]
        if with_grad:
            imgs, _ = rasterize_views(L["means3D"], L["means2D"], L["opacities"], sts, shs=L["shs"], scales=L["scales"],
                                      rotations=L["rotations"])
            (imgs * G).sum().backward()
            for t in L.values():
                t.grad = None
            return imgs.detach()
        with torch.no_grad():
            imgs, _ = rasterize_views(L["means3D"], L["means2D"], L["opacities"], sts, shs=L["shs"], scales=L["scales"],
                                      rotations=L["rotations"])
        return imgs

    timing_gather = [False]

    def gather(imgs):
        """imgs [n,3,H,W], n <= VPC: this rank's frames of one submission -> rank 0, on the gather stream."""
        n = imgs.shape[0]
        cur = torch.cuda.current_stream(dev)
        gather_stream.wait_stream(cur)               # run_frames_pipelined made `cur` wait for the submission's last kernel
        with torch.cuda.stream(gather_stream):
            if timing_gather[0]:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(gather_stream)
            src = imgs.cpu() if host_collectives else imgs.contiguous()
            src.record_stream(gather_stream) if src.is_cuda else None
            for w in multiview.gather_to_root(src, [b[:n] for b in gather_bufs] if rank == 0 else None, dst=0, mode=args.gather_mode):
                w.wait()            # (RCCL: orders the gather stream behind the transfer, the host does not block)
            if timing_gather[0]:
                e1.record(gather_stream)
                gather_events.append((e0, e1))

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(first, count, streams=None, vpc=None, with_grad=None, gather_on=None):
        """`count` steps from global step `first`, submitted --views-per-call at a time, --streams submissions in flight
        (pcrender.multiview.run_frames_pipelined); the frame gather is issued by this thread only, in step order, so every
        rank enqueues collectives identically."""
        wg = grad if with_grad is None else with_grad
        ch = []
        i = first
        while i < first + count:
            n = min(VPC if vpc is None else vpc, first + count - i)
            ch.append((i, n))
            i += n
        go = do_gather if gather_on is None else gather_on
        multiview.run_frames_pipelined(lambda ci, slot: render_many(ch[ci][0], ch[ci][1], slot, wg), 0, len(ch),
                                       args.streams if streams is None else streams,
                                       on_frame=(lambda ci, imgs: gather(imgs)) if go else None, device=dev)

    def timed(count, **kw):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(warm, count, gather_on=False, **kw)
        torch.cuda.synchronize()
        return time.perf_counter() - t1

    def stage_pass(count, **kw):
        """single-stream pass with hipEvents around every stage; returns ({stage: mean ms per launch}, wall seconds)"""
        torch.cuda.synchronize()
        _native.set_profiling(True)
        ovl_was = _native._OVERLAP_ON
        _native.set_overlap(False)    # per-stage times are taken in stream order: one kernel on the device at a time
        sub_tag[0] = "stage"          # (the shader-clock probes run beside this pass too)
        t1 = time.perf_counter()
        run_steps(warm, count, streams=1, gather_on=False, **kw)
        torch.cuda.synchronize()
        d1 = time.perf_counter() - t1
        sub_tag[0] = None
        sub_events.clear()
        _native.set_overlap(ovl_was)
        ms = {}
        for name, t in _native.get_profile():
            ms.setdefault(name, []).append(t)
        _native.set_profiling(False)
        return {k: float(np.mean(v)) for k, v in ms.items()}, d1

    def drop_in_literal(blocks=3, frames=48):
        """frames/s of the literal per-view loop: one thread, one stream, in order; median of `blocks` blocks after an untimed one"""
        was = _native._OVERLAP_ON
        _native.set_overlap(False)
        literal_mode[0] = True
        try:
            run_steps(0, frames, streams=1, vpc=1, gather_on=False)
            ts = [timed(frames, streams=1, vpc=1) for _ in range(blocks)]
        finally:
            literal_mode[0] = False
            _native.set_overlap(was)
        return frames / float(np.median(ts)), [round(frames / t, 1) for t in ts]

    def drop_in_host_exposed(frames=96):
        """Host time between the caller's last blocking copy (the settings are built) and the library's C entry point, which launches
        the frame's first kernel at once: what the GPU idles through in every frame of the literal loop (the copies drain the stream).
        Stamps around the real code path (the ctypes entry is wrapped for these frames only); scripts/literal_host_trace.py has the
        layer-by-layer version."""
        nonlocal host_trace
        real = _native.lib.gsr_forward_batch
        stamps = ([], [])

        def stamped(*a):
            stamps[1].append(time.perf_counter())
            return real(*a)
        was = _native._OVERLAP_ON
        _native.set_overlap(False)
        literal_mode[0] = True
        try:
            run_steps(0, 12, streams=1, vpc=1, gather_on=False)
            torch.cuda.synchronize()
            _native.lib.gsr_forward_batch, host_trace = stamped, stamps
            run_steps(0, frames, streams=1, vpc=1, gather_on=False)
            torch.cuda.synchronize()
        finally:
            _native.lib.gsr_forward_batch, host_trace = real, None
            literal_mode[0] = False
            _native.set_overlap(was)
        d = np.array(stamps[1][:len(stamps[0])]) - np.array(stamps[0][:len(stamps[1])])
        return {"mean": round(float(d.mean()) * 1e6, 1), "median": round(float(np.median(d)) * 1e6, 1), "frames": int(d.size),
                "what": "settings built -> gsr_forward_batch entered (the entry launches the first kernel at once), literal per-view loop"}

    if args.gather_probe:
        # World-size-1 anchor for the multi-GPU runs (nobody has measured more than one GPU yet): the same timed blocks with and
        # without the RCCL gather of every submission's full-size frames on its side stream, in ONE process group of one rank.
        # Alternating blocks; each block ends with the gather stream drained.
        def block(gather_on):
            fence()
            t1 = time.perf_counter()
            run_steps(0, args.steps, gather_on=gather_on)
            torch.cuda.synchronize()
            return time.perf_counter() - t1
        for _ in range(3):
            block(True)
        on, off = [], []
        for _ in range(max(3, args.repeats)):
            off.append(block(False))
            on.append(block(True))
        f_on, f_off = args.steps / float(np.median(on)), args.steps / float(np.median(off))
        print(json.dumps({"frames_per_s_with_gather": round(f_on, 1), "frames_per_s_without": round(f_off, 1),
                          "gather_overhead_pct": round(100.0 * (f_off / f_on - 1.0), 2), "backend": args.dist_backend,
                          "mode": args.gather_mode, "frame_bytes": 3 * W * H * 4, "views_per_call": VPC, "world": world}))
        if use_dist:
            dist.destroy_process_group()
        return

    if args.drop_in_probe:
        # a fresh process measuring only the drop-in figure (spawned by the main run, --drop-in-processes)
        warm = 0
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < args.warmup_seconds:
            run_steps(0, 24, streams=1, vpc=1, gather_on=False)
            torch.cuda.synchronize()
        fps, blocks = drop_in_literal()
        print(json.dumps({"drop_in_frames_per_s": round(fps, 1), "blocks": blocks}))
        return

    # untimed: the W warm-up steps asked for, plus priming of every stream's allocator pool / code objects
    # (3 frames per stream and 3 on the caller's stream) so that a small W does not put first-touch costs in the timing,
    # plus frames until --warmup-seconds have passed: the chip settles at the clock it sustains under this load
    warm = max(args.warmup, 3 * max(1, args.streams))
    t_w = time.perf_counter()
    run_steps(0, max(warm, VPC * max(1, args.streams) * 2), streams=1)      # primes the caller's stream and allocator pool
    run_steps(0, max(warm, VPC * max(1, args.streams) * 2))
    warm_steps = 2 * max(warm, VPC * max(1, args.streams) * 2)
    torch.cuda.synchronize()
    more = torch.zeros(1, dtype=torch.int32)
    while True:
        # every rank must run the same number of warm-up rounds (they contain collectives): rank 0's clock decides
        more[0] = int(time.perf_counter() - t_w < args.warmup_seconds)
        if use_dist:
            mt = more.to("cpu" if host_collectives else dev)
            dist.broadcast(mt, src=0)
            more.copy_(mt)
        if not int(more[0]):
            break
        run_steps(0, max(args.steps, VPC))
        warm_steps += max(args.steps, VPC)
        torch.cuda.synchronize()
    # first use of the in-region instrumentation happens HERE, not in the first timed block: the probe kernel's code object is
    # loaded on its first launch and torch grows its event pool (7 ms in the first block otherwise)
    sub_tag[0] = "warm"
    run_steps(0, max(args.steps, VPC))
    sub_tag[0] = None
    warm_steps += max(args.steps, VPC)
    torch.cuda.synchronize()
    sub_events.clear()
    warm_seconds = time.perf_counter() - t_w
    fence()
    # Python's cyclic garbage collector is kept out of the timed region, as timeit does: a generation-2 collection is a 30 ms host
    # stall (it landed in the second submission of the third block of every --steps 20 run: 34 ms instead of 4.6); collected now
    import gc
    gc.collect()
    gc.disable()
    # the collection left the GPU idle for tens of ms and the clock governor let go: one 20-step block (10 ms) does not bring the
    # clock back -- the first timed block of a --steps 20 run read 2 290-2 370 MHz, the second 2 380-2 420, the rest 2 430 -- so
    # untimed blocks run until GC_RECOVER_S of load have passed (rank 0's clock decides, like the warm-up above)
    t_g = time.perf_counter()
    while True:
        run_steps(0, max(args.steps, VPC))
        warm_steps += max(args.steps, VPC)
        torch.cuda.synchronize()
        more[0] = int(time.perf_counter() - t_g < GC_RECOVER_S)
        if use_dist:
            mt = more.to("cpu" if host_collectives else dev)
            dist.broadcast(mt, src=0)
            more.copy_(mt)
        if not int(more[0]):
            break
    warm_seconds = time.perf_counter() - t_w
    fence()
    block_dt = []
    block_base = []          # one event per timed block, recorded on the caller's stream when the block starts
    tail_events = []         # per block: (last render work done, last gather done) -> the gather time nothing hides
    nxt = warm
    timing_gather[0] = True
    probe_first = probe.n
    probe_block = []
    _native.OVERLAP_STATS.update(calls=0, overlapped=0)
    # the two render kernels are timed INSIDE the timed region (one hipEvent pair per forward and per backward on the launch
    # stream): `roofline.avg_ms` is the dominant kernel's duration among the very launches `value` counts
    if not args.no_stage_events:
        _native.set_profiling(2)
    for b in range(max(1, args.repeats)):
        base = torch.cuda.Event(enable_timing=True)
        base.record(torch.cuda.current_stream(dev))
        block_base.append(base)
        sub_tag[0] = b
        probe_block.append(probe.n)
        t0 = time.perf_counter()
        run_steps(nxt, args.steps)
        sub_tag[0] = None
        if do_gather:
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record(torch.cuda.current_stream(dev))       # behind every submission of the block (run_steps joined its workers)
            eb.record(gather_stream)
            tail_events.append((ea, eb))
        fence()
        block_dt.append(time.perf_counter() - t0)
        nxt += args.steps
    timing_gather[0] = False
    inreg = {}
    if not args.no_stage_events:
        for name, t in _native.get_profile():
            inreg.setdefault(name, []).append(t)
        _native.set_profiling(0)
    # (a timed block whose step count is not a multiple of --views-per-call ends with a shorter launch: only full launches count)
    full_launches = (args.steps // VPC) * max(1, args.repeats)
    inreg_ms = {k: float(np.mean(sorted(v, reverse=True)[:max(1, min(len(v), full_launches))])) if args.steps % VPC else float(np.mean(v))
                for k, v in inreg.items()}
    overlap_timed = "%d of %d" % (_native.OVERLAP_STATS["overlapped"], _native.OVERLAP_STATS["calls"])
    gc.enable()
    sclk_timed = probe.mhz(probe_first)
    probe_block.append(probe.n)
    sclk_blocks = [round(float(np.median(sclk_timed[a - probe_first:b - probe_first])), 1) if b > a else None
                   for a, b in zip(probe_block[:-1], probe_block[1:])] if sclk_timed else None
    gather_ms = [float(a.elapsed_time(b)) for a, b in gather_events] if gather_events else None
    gather_exposed_ms = [max(0.0, float(a.elapsed_time(b))) for a, b in tail_events] if tail_events else None
    # GPU time of the timed steps: the union of the submissions' [start, end] intervals on the device clock, per block
    gpu_busy_ms, sub_ms = [], []
    for b in range(len(block_dt)):
        iv = [(float(block_base[b].elapsed_time(e0)), float(block_base[b].elapsed_time(e1))) for tag, e0, e1 in sub_events if tag == b]
        gpu_busy_ms.append(union_ms(iv))
        sub_ms.append([round(b1 - a1, 3) for a1, b1 in sorted(iv)])
    sub_events.clear()
    block_dt_local = list(block_dt)

    avg_ms, single = {}, None
    sclk_stage = []
    per_view = drop_in = fwd_only = rgb_time = None
    if rank == 0 and not args.no_stage_events:
        # whole --views-per-call submissions only: `kernels_ms` is a mean per LAUNCH and is divided by the views of a launch below
        # (a pass that mixed 12- and 8-view launches understated every per-frame figure by 17 %: the driver's --steps 20 run of
        # round 3 and its "18 % faster kernels")
        n1 = 2 * VPC
        probe_stage = probe.n
        avg_ms, d1 = stage_pass(n1)
        sclk_stage = probe.mhz(probe_stage)
        single = {"frames_per_s": round(n1 / d1, 3), "ms_per_frame": round(d1 / n1 * 1e3, 4), "frames": n1,
                  "note": "one submission in flight, hipEvents around every stage (each pair costs ~10 us of bubble)"}
    if rank == 0 and not args.no_per_view and not args.no_stage_events:
        # ---- the drop-in API: the reference's call pattern, one GaussianRasterizer call per view (simple_raw_render.py:259-278)
        drop_in = {"call": "GaussianRasterizer(settings_v)(means3D, means2D, opacities, shs=, scales=, rotations=) per view + "
                           "loss.backward() per view" if grad else "GaussianRasterizer(...) per view under no_grad",
                   "frames_per_s": {}}
        # literal: the reference caller's loop body as it stands -- settings built per call inside the timed loop (drop_in_literal);
        # one_stream_in_order: prebuilt GaussianRasterizer objects, the library's default (plain stream order);
        # one_stream_overlap_opt_in: the same with GSR_OVERLAP=1 (a view's front end starts beside the previous view's backward while
        # every input tensor is provably unchanged; what it is worth depends on the hardware queues the runtime hands out, which is
        # why it is off by default); four_streams: four caller threads, each with its own stream and leaf tensors
        fps_lit, lit_blocks = drop_in_literal()
        drop_in["frames_per_s"]["literal"] = round(fps_lit, 1)
        drop_in["literal_blocks"] = lit_blocks
        drop_in["host_exposed_us"] = drop_in_host_exposed()
        ovl_default = _native._OVERLAP_ON
        for name, st, ovl in (("one_stream_in_order", 1, False), ("one_stream_overlap_opt_in", 1, True), ("four_streams", 4, False)):
            if st > len(leafsets):
                continue
            _native.set_overlap(ovl)
            _native.OVERLAP_STATS.update(calls=0, overlapped=0)
            # every worker stream has to grow its own allocator pool and arenas first: 48 untimed frames, then the median of
            # three blocks of 48 (a single cold block of 48 frames reported 410-700 frames/s for what runs at 1 400)
            run_steps(warm, 48, streams=st, vpc=1, gather_on=False)
            drop_in["frames_per_s"][name] = round(48 / float(np.median([timed(48, streams=st, vpc=1) for _ in range(3)])), 1)
            if name == "one_stream_overlap_opt_in":
                drop_in["overlapped_calls"] = "%d of %d" % (_native.OVERLAP_STATS["overlapped"], _native.OVERLAP_STATS["calls"])
        _native.set_overlap(ovl_default)
        pv_ms, _ = stage_pass(24, vpc=1)
        drop_in["kernels_ms_per_frame"] = {k: round(v, 4) for k, v in pv_ms.items()}
        drop_in["kernel_sum_ms_per_frame"] = round(sum(pv_ms.values()), 4)
        per_view = drop_in["frames_per_s"]
        # ---- inference (forward only), both call shapes
        if grad:
            run_steps(warm, 2 * VPC, with_grad=False, gather_on=False)
            med3 = lambda n, **kw: float(np.median([timed(n, **kw) for _ in range(3)]))   # noqa: E731
            fwd_only = {"views_per_call_%d_frames_per_s" % VPC: round(4 * VPC / med3(4 * VPC, with_grad=False), 1),
                        "per_view_call_frames_per_s": round(48 / med3(48, streams=1, vpc=1, with_grad=False), 1)}
        # ---- the reference's own timing hook: `rgb time` = 12 views x (512^2 camera, super-sample 2 -> 1024^2 raster), SH pass,
        # per-view settings construction and the bilinear down-filter INSIDE the timed region (simple_raw_render.py:433-456)
        try:
            from pcrender import raster_passes as rp
            Hs = camera.circle_path(12, 0, 3, [90, 0]).unsqueeze(0)
            sf = float(cloud["scale_factor"])
            radius = float(np.sqrt(3) / sf * 6)
            dec_s = (scales.detach() / radius).contiguous()
            with torch.no_grad():
                def rgb_pass(batch_views):
                    return rp.rasterize_views([means3D.detach()], [opac.detach()], [dec_s], [rots.detach()], Hs, 512, 512, 45.0,
                                              torch.ones(3), sf, shs_list=[shs.detach()], sh_degree=D, batch_views=batch_views)
                rgb_time = {"what": "12 circle views, 512x512 camera x super-sample 2 (1024x1024 raster), SH colour pass, per-view "
                                    "settings glue + bilinear down-filter included; this workload's cloud", "ms_per_12_views": {}}
                for name, bv in (("literal_per_view_calls", False), ("views_in_one_call", True)):
                    for _ in range(2):
                        rgb_pass(bv)
                    its = []
                    for _ in range(5):
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        rgb_pass(bv)
                        torch.cuda.synchronize()
                        its.append((time.perf_counter() - t1) * 1e3)
                    rgb_time["ms_per_12_views"][name] = round(float(np.median(its)), 3)
                    rgb_time.setdefault("iterations_ms", {})[name] = [round(x, 2) for x in its]
        except Exception as ex:  # noqa: BLE001 -- a side figure must never take the headline down
            rgb_time = {"error": repr(ex)}
    if drop_in is not None and args.drop_in_processes > 0 and world == 1:
        # the drop-in figure from FRESH processes (each: import, build the cloud, one second of warm-up, three blocks of 48 frames):
        # it must not depend on what this process did before, nor on how a process's streams happened to be set up
        torch.cuda.synchronize()
        drop_in["fresh_processes"] = _dist.drop_in_fresh_processes(args, os.path.abspath(__file__), W, H)
    gather_anchor = None
    if world == 1 and not use_dist and not args.no_per_view and not args.no_gather:
        torch.cuda.synchronize()
        gather_anchor = _dist.gather_anchor(args, os.path.abspath(__file__), W, H, dev_index)
    per_rank_blocks = None
    if use_dist:
        cdev = "cpu" if host_collectives else dev
        t = torch.tensor(block_dt, device=cdev, dtype=torch.float64)
        every = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(every, t)                            # each rank's own block times (before the MAX): per-rank frames/s
        per_rank_blocks = [[float(x) for x in e.tolist()] for e in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        block_dt = [float(x) for x in t.tolist()]
        # where every rank ran: (rank, local rank, device index, host threads), two int32 per field through the same backend
        mine = torch.tensor([rank, local_rank, dev_index, torch.get_num_threads()], device=cdev, dtype=torch.int32)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        devs = [dict(zip(("rank", "local_rank", "device", "host_threads"), (int(x) for x in r.tolist()))) for r in allr]
    dt = float(np.median(block_dt))

    if rank == 0:
        # ---- workload statistics for the bytes model, averaged over the views rank 0 rendered
        used = sorted({view_of(warm + i, 0, world, n_views, shard) for i in range(args.steps * max(1, args.repeats))})
        stats = dict(V=0.0, R=0.0, L=0.0, C_fwd=0.0, C_bwd=0.0)
        T = ((W + 15) // 16) * ((H + 15) // 16)
        with torch.no_grad():
            for c0 in range(0, len(used), VPC):
                vs = used[c0:c0 + VPC]
                vm = torch.stack([settings[v].viewmatrix.reshape(4, 4) for v in vs])
                pm = torch.stack([settings[v].projmatrix.reshape(4, 4) for v in vs])
                cp = torch.stack([settings[v].campos.reshape(3) for v in vs])
                s0 = settings[vs[0]]
                counts, _, radii, geom, binning, img = _native.rasterize_gaussians_batch(
                    s0.bg, means3D.detach(), torch.empty(0), opac.detach(), scales.detach(), rots.detach(), 1.0, torch.empty(0), vm, pm,
                    s0.tanfovx, s0.tanfovy, H, W, shs.detach(), D, cp, False, False, need_backward=True)
                for k in range(len(vs)):
                    R = counts[k]
                    need = _native.query("TILE_NEED", P, W, H, R, geom, binning, img, view=k, n_views=len(vs)).long()
                    nc = _native.query("N_CONTRIB", P, W, H, R, geom, binning, img, view=k, n_views=len(vs)).view(H, W)
                    ncp = torch.zeros((((H + 15) // 16) * 16, ((W + 15) // 16) * 16), dtype=nc.dtype, device=dev)
                    ncp[:H, :W] = nc
                    cb = ncp.view(ncp.shape[0] // 16, 16, ncp.shape[1] // 16, 16).amax(dim=(1, 3)).long().sum()
                    stats["V"] += float((radii[k] > 0).sum()) / len(used)
                    stats["R"] += float(R) / len(used)
                    stats["L"] += float(_native.query("LIST_PAIRS", P, W, H, R, geom, binning, img, view=k, n_views=len(vs))[0]) / len(used)
                    stats["C_fwd"] += float(need.sum()) / len(used)
                    stats["C_bwd"] += float(cb) / len(used)
        tile_bits = int(T).bit_length()
        bytes_per = algorithmic_bytes(P, stats["V"], stats["L"], T, W * H, (D + 1) ** 2, stats["C_fwd"], stats["C_bwd"],
                                      (tile_bits + 7) // 8, views_per_call=VPC)
        roofline = _roofline.build(avg_ms, inreg_ms, inreg, bytes_per, VPC, args, P, W, H, sclk_timed, sclk_stage)
        frame_bytes = sum(bytes_per[k] for k in bytes_per if (grad or "backward" not in k))
        frame_gpu_ms = sum(avg_ms.values()) / VPC if avg_ms else 0.0

        cpu = _cpu_baseline.measure(views, g, W, H, D, G, grad, args) if (world == 1 and not args.no_cpu_baseline) else None
        ref_ctx = None
        try:
            with open(os.path.join(ROOT, "profiles", "reference_build.json")) as f:
                ref_ctx = json.load(f)
        except (OSError, ValueError):
            pass
        shape = ("%d views per rasterize_views call (C ABI gsr_forward_batch%s), %d call%s in flight per rank" % (
            VPC, " / gsr_backward_batch" if grad else "", args.streams, "s" if args.streams != 1 else "")) if VPC > 1 else (
            "one GaussianRasterizer call per view (the reference's call pattern), %d in flight per rank" % args.streams)
        out = {
            "metric": "rendered frames/sec at 1080p (fwd+bwd), THuman-800K" if (grad and (W, H) == (1920, 1080)) else
                      "rendered frames/sec %dx%d (%s)" % (W, H, "fwd+bwd" if grad else "fwd"),
            "value": round(world * args.steps / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "repeats": len(block_dt),
            "ms_per_step_blocks": [round(x / args.steps * 1e3, 4) for x in block_dt],
            "warmup": args.warmup,
            "warmup_effective": {"steps": int(warm_steps), "seconds": round(warm_seconds, 3),
                                 "note": "--warmup steps, allocator / stream priming, then whole blocks until --warmup-seconds "
                                         "(%.2f) had passed, the garbage collection, then whole blocks for %.2f s more (the collection "
                                         "idles the GPU for tens of ms and the clock needs ~50 ms of load to come back); all untimed"
                                         % (args.warmup_seconds, GC_RECOVER_S)},
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "python_gc": "collected before, disabled inside the timed region (like timeit)",
            "overlap_of_consecutive_calls": {"on": bool(_native._OVERLAP_ON), "calls_overlapped_in_timed_region": overlap_timed,
                                             "note": "opt-in (GSR_OVERLAP=1): the library then starts a call's front end beside the previous "
                                                     "call's backward when every input tensor is provably unchanged (_native._OnSideStream); "
                                                     "off by default: every kernel runs in the caller's stream order"},
            # GPU time of the TIMED steps themselves (rank 0): per block, the union of the submissions' [start, end] hipEvent
            # intervals on the streams they ran on (loss kernels included), and the block's wall time over it
            "gpu_ms_per_step_timed": round(float(np.median(gpu_busy_ms)) / args.steps, 4) if gpu_busy_ms else None,
            "gpu_ms_per_step_timed_blocks": [round(x / args.steps, 4) for x in gpu_busy_ms],
            "gpu_ms_submissions_blocks": sub_ms if sum(len(x) for x in sub_ms) <= 64 else None,   # per block, every submission's own span
            "wall_over_gpu": round(float(np.median([w * 1e3 / g for w, g in zip(block_dt_local, gpu_busy_ms) if g > 0])), 4)
            if gpu_busy_ms and all(g > 0 for g in gpu_busy_ms) else None,
            "sclk_mhz": {"timed_region": round(float(np.median(sclk_timed)), 1) if sclk_timed else None,
                         "timed_region_min_max": [round(min(sclk_timed), 1), round(max(sclk_timed), 1)] if sclk_timed else None,
                         "timed_blocks": sclk_blocks,
                         "stage_pass": round(float(np.median(sclk_stage)), 1) if sclk_stage else None,
                         "probes": len(sclk_timed) + len(sclk_stage),
                         "how": "one-wave probe kernel per submission on a side stream, beside the frame's kernels: s_memtime "
                                "(shader cycles) over s_memrealtime (%d kHz)" % probe.khz},
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d %s, %d circle views, %s profile, SH degree %d (M=%d); %s; %s%s" % (
                args.workload, W, H, "fwd+bwd" if grad else "fwd", n_views, args.profile, D, M, shape,
                "every rank walks the circle, rank r starting (r x %d) // world views in" % n_views if shard == "circle" else
                "the %d views of a turn dealt round-robin to the ranks (rank r owns v mod world == r)" % n_views,
                " + RCCL frame gather on its own stream" if do_gather else ""),
                "baseline_config": cfg["what"],
                "call_shape": {"views_per_call": VPC, "calls_in_flight": args.streams,
                               "entry_point": "rasterize_views" if VPC > 1 else "GaussianRasterizer.forward"},
                "points": P, "num_rendered_avg": int(stats["R"]), "list_pairs_avg": int(stats["L"]),
                "list_pairs_note": "num_rendered = pairs of the reference's tile rectangles (reported bit-exact); list_pairs = pairs the "
                                   "library emits and sorts: rectangles clipped to where alpha can reach 1/255 (gsr_params.reference_lists = 0, "
                                   "the default; GSR_REFERENCE_LISTS=1 restores the reference's lists)",
                "visible_avg": int(stats["V"]),
                "consumed_entries_fwd_avg": int(stats["C_fwd"]), "consumed_entries_bwd_avg": int(stats["C_bwd"])},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "kernels_sha": kernels_sha(),
            # the reference's API as its caller uses it: per view, settings built inside the loop, one call, one backward; one thread,
            # one stream, in order (drop_in_api has the variants)
            "drop_in": None if drop_in is None else {
                "frames_per_s": drop_in["frames_per_s"].get("literal"), "unit": "frames/s",
                "fraction_of_value": round(drop_in["frames_per_s"].get("literal", 0.0) / (world * args.steps / dt), 4),
                "what": "the reference caller's loop body as it stands (simple_raw_render.py:260-278): per view the settings are built "
                        "for the call (fresh tensors), GaussianRasterizer(settings)(means3D, means2D, opacities, shs=, scales=, "
                        "rotations=), loss.backward(); ONE thread, ONE stream, plain stream order (the library's default)",
                "host_exposed_us": drop_in.get("host_exposed_us"),
                "fresh_processes": drop_in.get("fresh_processes")},
            "reference_build_context": ref_ctx,
            "multi_gpu_status": "measured on %d GPU(s)" % world if (world > 1 and not host_collectives) else
                                "no run on more than one GPU has been measured (one-GPU leases only): the N > 1 path is rehearsed with "
                                "gloo ranks sharing one GPU (tests/test_gpu_bench.py) and 8 CPU ranks (tests/test_cpu_multiview.py)",
            # world-size-1 anchor of the frame gather (a child process with a 1-rank RCCL group): frames/s with the gather of every
            # submission's frames on its side stream vs without -- what the first measured 8-GPU run starts from
            "gather_world1_anchor": gather_anchor,
            "kernels_ms": {k: round(v, 4) for k, v in avg_ms.items()},
            "kernel_timing": "hipEvents on the launch stream, single-stream pass right after the timed region",
            "views_per_call": VPC, "kernels_ms_per_frame": {k: round(v / VPC, 4) for k, v in avg_ms.items()},
            # per stage: algorithmic bytes of a launch (the per-call-shape model of algorithmic_bytes) over its measured duration
            "kernels_algorithmic_GBps": {k: round(bytes_per[k] * VPC / (v * 1e-3) / 1e9, 1) for k, v in avg_ms.items()
                                         if k in bytes_per and v > 0},
            "streams_per_rank": args.streams, "single_stream": single,
            "drop_in_api": drop_in, "per_view_api_frames_per_s": per_view,
            "forward_only": fwd_only, "rgb_time_equiv": rgb_time,
            "frame_hbm": {"algorithmic_bytes": int(frame_bytes), "gpu_ms_sum": round(frame_gpu_ms, 4),
                          "GBps": round(frame_bytes / (frame_gpu_ms * 1e-3) / 1e9, 1) if frame_gpu_ms else None,
                          "frac_of_8000": round(frame_bytes / (frame_gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if frame_gpu_ms else None,
                          "frac_of_6300": round(frame_bytes / (frame_gpu_ms * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBS, 4) if frame_gpu_ms else None},
        }
        if use_dist:
            out["per_rank_frames_per_s"] = [round(args.steps / float(np.median(b)), 1) for b in per_rank_blocks]
            out["distributed"] = {"world_size": dist.get_world_size(), "world": dist.get_world_size(), "backend": dist.get_backend(),
                                  "ranks": devs,
                                  "gather": None if not gather_ms else {
                                      "mode": args.gather_mode,
                                      "exposed_ms_per_block": [round(x, 4) for x in gather_exposed_ms] if gather_exposed_ms else None,
                                      "exposed_note": "hipEvent time from the end of a block's last submission to the end of its last "
                                                      "gather on rank 0: the part of the gather no render kernel hides",
                                      "collectives": len(gather_ms), "bytes_into_rank0_per_collective": int((world - 1) * VPC * 3 * H * W * 4),
                                      "ms_mean": round(float(np.mean(gather_ms)), 4), "ms_max": round(float(np.max(gather_ms)), 4),
                                      "note": "hipEvents on the gather stream around each dist.gather (includes waiting for the peers' "
                                              "submissions); the gather overlaps the next submission's kernels"}}
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
