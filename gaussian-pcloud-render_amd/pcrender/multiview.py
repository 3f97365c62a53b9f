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


def gather_frames(local_frames, n_views, dst=0, group=None):
    """Gather per-rank frame stacks to `dst`.

    local_frames: [k_rank, 3, H, W] for the views shard_views(n_views, rank, world) in that order.
    Returns on dst a [n_views, 3, H, W] tensor ordered by view id; None elsewhere.  Shards of unequal length
    (n_views % world != 0) are padded to the longest shard for the collective and trimmed afterwards.
    """
    if not (dist.is_available() and dist.is_initialized()):
        return local_frames
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    k = max_shard(n_views, world)
    shape = tuple(local_frames.shape[1:])
    pad = torch.zeros((k,) + shape, dtype=local_frames.dtype, device=local_frames.device)
    pad[: local_frames.shape[0]] = local_frames
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, gather_list=bufs, dst=dst, group=group)
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


def render_views(render_one, n_views, dst=0, group=None):
    """render_one(view_id) -> [3,H,W] tensor.  Renders this rank's shard, gathers every frame on `dst`."""
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    frames = [render_one(v) for v in shard_views(n_views, rank, world)]
    if not frames:
        raise ValueError("rank %d owns no view (n_views=%d < world=%d)" % (rank, n_views, world))
    return gather_frames(torch.stack(frames, 0), n_views, dst=dst, group=group)
