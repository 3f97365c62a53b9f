"""View-sharded rendering across the GPUs of one node: one process per GPU, torch.distributed.

Camera views are independent units of work (the reference already loops over them sequentially,
/root/reference/simple_raw_render.py:259-278), so they shard with no data-path exchange: rank g renders
views {v : v mod world == g} of the same (replicated) Gaussian cloud.  The only collective is the gather
of finished frames to rank 0 -- [3,H,W] fp32 = 24.9 MB per 1080p frame, peer -> root over direct xGMI links
(backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests of this module).
"""
import torch
import torch.distributed as dist


def shard_views(n_views, rank, world):
    """View ids owned by `rank` (round-robin keeps neighbouring, similarly expensive views on different GPUs)."""
    return list(range(rank, n_views, world))


def max_shard(n_views, world):
    return (n_views + world - 1) // world


def _global_rank(group, r):
    """torch.distributed's P2POp peers and collective roots are GLOBAL ranks; `r` counts inside `group`"""
    return r if group is None else dist.get_global_rank(group, r)


def gather_to_root(src, bufs, dst=0, group=None, mode="collective"):
    """One frame-gather step: every rank's `src` lands in bufs[rank] on `dst` (bufs: list of world tensors shaped like src on dst,
    None elsewhere).  `dst` and the indices of `bufs` are ranks WITHIN `group` (the default group: global ranks).

    mode "collective": torch.distributed.gather -- one RCCL collective; how it drives the links is the library's business.
    mode "p2p": the root posts one receive per peer and every peer one send, all in ONE batch_isend_irecv group
        (ncclGroupStart / ncclRecv x (world - 1) / ncclGroupEnd on RCCL): xGMI is point-to-point, seven direct links end at the root,
        and a grouped set of receives keeps all of them busy at once even if the collective were implemented as a serial loop
        over the peers.  The fallback DESIGN.md section 8 names; same result, same call order on every rank.
    Returns the list of outstanding work handles (empty for "collective"): wait() on them before reading bufs / reusing src."""
    if mode == "collective":
        dist.gather(src, gather_list=bufs, dst=_global_rank(group, dst), group=group)
        return []
    if mode != "p2p":
        raise ValueError("gather mode: 'collective' or 'p2p'")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if rank == dst:
        bufs[dst].copy_(src)
        ops = [dist.P2POp(dist.irecv, bufs[r], _global_rank(group, r), group) for r in range(world) if r != dst]
    else:
        ops = [dist.P2POp(dist.isend, src, _global_rank(group, dst), group)]
    return dist.batch_isend_irecv(ops) if ops else []


def gather_frames(local_frames, n_views, dst=0, group=None, mode="collective"):
    """Gather per-rank frame stacks to `dst`.

    local_frames: [k_rank, 3, H, W] for the views shard_views(n_views, rank, world) in that order.
    Returns on dst a [n_views, 3, H, W] tensor ordered by view id; None elsewhere.  Shards of unequal length
    (n_views % world != 0) are padded to the longest shard for the collective and trimmed afterwards.
    mode: see gather_to_root.
    """
    if not (dist.is_available() and dist.is_initialized()):
        return local_frames
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    k = max_shard(n_views, world)
    shape = tuple(local_frames.shape[1:])
    pad = torch.zeros((k,) + shape, dtype=local_frames.dtype, device=local_frames.device)
    pad[: local_frames.shape[0]] = local_frames
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    for w in gather_to_root(pad, bufs, dst=dst, group=group, mode=mode):
        w.wait()
    if rank != dst:
        return None
    out = torch.empty((n_views,) + shape, dtype=local_frames.dtype, device=local_frames.device)
    for r in range(world):
        ids = shard_views(n_views, r, world)
        if ids:
            out[ids] = bufs[r][: len(ids)]
    return out


def reduce_gradients(grads, group=None):
    """Sum per-Gaussian gradients of a shared cloud over ranks (views of one loss live on different GPUs)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for g in grads:
            if g is not None:
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    return grads


def render_views(render_one, n_views, dst=0, group=None, mode="collective"):
    """render_one(view_id) -> [3,H,W] tensor.  Renders this rank's shard, gathers every frame on `dst`."""
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    frames = [render_one(v) for v in shard_views(n_views, rank, world)]
    if not frames:
        raise ValueError("rank %d owns no view (n_views=%d < world=%d)" % (rank, n_views, world))
    return gather_frames(torch.stack(frames, 0), n_views, dst=dst, group=group, mode=mode)


_STREAMS = {}


def _worker_streams(device, n):
    """The worker streams are kept for the life of the process: torch's caching allocator pools memory per stream, so
    fresh streams on every call would start with empty pools and pay hipMalloc (a device-wide sync) for every arena of
    their first frames."""
    key = (str(torch.device(device)), n)
    if key not in _STREAMS:
        _STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _STREAMS[key]


def run_frames_pipelined(render, first, count, n_inflight, on_frame=None, device=None):
    """Keep `n_inflight` independent frames in flight on one GPU.

    `render(step, slot) -> tensor` renders one whole frame (forward, and backward if the caller wants) on the
    calling thread's current stream; `slot` in [0, n_inflight) names the worker so the caller can give every worker
    its own leaf tensors.  Frames are independent (views of one cloud), so `n_inflight` host threads each render
    every n_inflight-th step on their own HIP stream: one frame's tail overlaps the next frame's bandwidth-bound
    stages (ctypes releases the GIL during the native calls, including the one stream sync per frame).

    Every worker stream should get a hardware queue of its own: start the process with GPU_MAX_HW_QUEUES >= n_inflight + 2
    (the ROCm default of 4 makes two of four workers' streams, or a worker and the caller's stream, share a queue and
    serialise; bench.py sets 8).

    `on_frame(step, tensor)`, if given, is called on the CALLING thread, in increasing step order, after making the
    caller's stream wait for the frame - this is where collectives (the frame gather) belong, so that every rank
    issues them in the same order whatever the thread timing.  With device=None (CPU tests) no streams are used.
    """
    import threading

    if n_inflight <= 1:
        for i in range(first, first + count):
            img = render(i, 0)
            if on_frame is not None:
                on_frame(i, img)
        return
    use_cuda = device is not None and torch.device(device).type == "cuda"
    streams = _worker_streams(device, n_inflight) if use_cuda else [None] * n_inflight
    errs = []
    ready = {i: threading.Event() for i in range(first, first + count)}
    frames = {}

    def worker(t):
        try:
            if use_cuda:
                torch.cuda.set_device(device)
            ctx = torch.cuda.stream(streams[t]) if use_cuda else _NullCtx()
            with ctx:
                for i in range(first + t, first + count, n_inflight):
                    img = render(i, t)
                    ev = None
                    if use_cuda:
                        ev = torch.cuda.Event()
                        ev.record(streams[t])
                    frames[i] = (img, ev)
                    ready[i].set()
            if use_cuda:
                streams[t].synchronize()
        except Exception as e:  # noqa: BLE001 - re-raised on the calling thread
            errs.append(e)
            for e2 in ready.values():
                e2.set()

    cur = torch.cuda.current_stream(device) if use_cuda else None
    if use_cuda:
        for st in streams:
            st.wait_stream(cur)
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_inflight)]
    for x in threads:
        x.start()
    for i in range(first, first + count):
        ready[i].wait()
        if errs:
            break
        img, ev = frames.pop(i)
        if on_frame is not None:
            if ev is not None:
                cur.wait_event(ev)
            on_frame(i, img)
    for x in threads:
        x.join()
    if errs:
        raise errs[0]


class _NullCtx(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
