"""CPU tests of the N>1 path: view sharding + frame gather + gradient reduction with torch.distributed `gloo`,
world_size 2 (and 3 for an uneven split), one process per rank, rendezvous on 127.0.0.1; an EIGHT-rank rehearsal of BASELINE
configs[3] / [4]'s shape (8 views, one per rank; no 8-GPU node has run this code yet); a gather inside a sub-group."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-pcloud-render_amd"))


def _frame(v, H=6, W=10):
    """Deterministic stand-in for a rendered frame of view v."""
    g = torch.Generator().manual_seed(1000 + v)
    return torch.rand((3, H, W), generator=g)


def _worker(rank, world, port, n_views, q, mode="collective"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pcrender import multiview
    try:
        out = multiview.render_views(_frame, n_views, dst=0, mode=mode)
        grads = [torch.full((4, 3), float(rank + 1)), None, torch.arange(5, dtype=torch.float32) * (rank + 1)]
        multiview.reduce_gradients(grads)
        ok = True
        if rank == 0:
            want = torch.stack([_frame(v) for v in range(n_views)], 0)
            ok = out is not None and out.shape == want.shape and torch.equal(out, want)
        else:
            ok = out is None
        tot = world * (world + 1) / 2
        ok = ok and torch.equal(grads[0], torch.full((4, 3), tot)) and grads[1] is None
        ok = ok and torch.equal(grads[2], torch.arange(5, dtype=torch.float32) * tot)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _run(world, n_views, port, mode="collective"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_views, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]


def test_shard_views_partition():
    from pcrender import multiview
    for n_views, world in [(12, 1), (12, 2), (12, 8), (8, 8), (12, 5)]:
        parts = [multiview.shard_views(n_views, r, world) for r in range(world)]
        assert sorted(v for p in parts for v in p) == list(range(n_views))
        assert max(len(p) for p in parts) == multiview.max_shard(n_views, world)
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_gather_world2_even():
    _run(2, 12, 29631)


def test_gather_world3_uneven_shards():
    _run(3, 8, 29632)     # shards of 3,3,2 views: padded for the collective, trimmed on the root


def test_gather_p2p_fallback_world2_and_world3():
    """the grouped send / receive gather (batch_isend_irecv: DESIGN.md section 8's fallback for a root gather that does not
    drive the seven xGMI links at once) delivers what the collective delivers, even and uneven shards"""
    _run(2, 12, 29636, mode="p2p")
    _run(3, 8, 29637, mode="p2p")


def test_eight_ranks_one_view_each_both_gather_modes():
    """BASELINE configs[3] / [4]: 8 views dealt to 8 ranks (rank r owns view r), frames gathered on rank 0, gradients all-reduced"""
    _run(8, 8, 29641)
    _run(8, 8, 29642, mode="p2p")


def test_eight_ranks_twelve_views_uneven():
    _run(8, 12, 29643)     # the headline's 12 circle views on 8 ranks: shards of 2,2,2,2,1,1,1,1


def _subgroup_worker(rank, world, port, q, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pcrender import multiview
    try:
        members = [1, 2, 3]                       # global ranks of the group = group ranks 0, 1, 2 (new_group sorts them)
        grp = dist.new_group(ranks=members)
        ok = True
        if rank in members:
            me = dist.get_rank(grp)
            ok = me == members.index(rank)
            src = _frame(10 + me)
            root = 1                              # GROUP rank 1 = global rank 2
            bufs = [torch.empty_like(src) for _ in members] if me == root else None
            for w in multiview.gather_to_root(src, bufs, dst=root, group=grp, mode=mode):
                w.wait()
            if me == root:
                ok = ok and all(torch.equal(bufs[r], _frame(10 + r)) for r in range(len(members)))
        dist.barrier()            # (the rank outside the group must not tear its connections down under the others' gather)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["collective", "p2p"])
def test_gather_inside_a_subgroup_uses_group_ranks(mode):
    """dst and the buffer indices count inside the group; the P2POp peers / the collective's root are translated to global ranks
    (group ranks 0, 1, 2 = global ranks 1, 2, 3; root = group rank 1 = global rank 2; global rank 0 stays outside)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29644 if mode == "collective" else 29645
    procs = [ctx.Process(target=_subgroup_worker, args=(r, 4, port, q, mode)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(4)]


def test_single_process_passthrough():
    from pcrender import multiview
    out = multiview.render_views(_frame, 4)
    assert out.shape == (4, 3, 6, 10) and torch.equal(out[2], _frame(2))
    with pytest.raises(ValueError):
        multiview.render_views(_frame, 0)


def _pipe_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pcrender import multiview
    try:
        import time
        got = []
        bufs = [torch.empty(3, 6, 10) for _ in range(world)] if rank == 0 else None

        def render(i, slot):
            time.sleep(0.002 * ((i * 7 + rank * 3 + slot) % 5))   # scramble thread timing
            return _frame(i * world + rank)

        def on_frame(i, img):
            dist.gather(img, gather_list=bufs, dst=0)
            if rank == 0:
                got.append((i, [b.clone() for b in bufs]))

        multiview.run_frames_pipelined(render, 3, 10, 4, on_frame=on_frame, device=None)
        ok = True
        if rank == 0:
            ok = [i for i, _ in got] == list(range(3, 13))
            for i, fr in got:
                for r in range(world):
                    ok = ok and torch.equal(fr[r], _frame(i * world + r))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_pipelined_frames_gather_in_step_order_world2():
    """4 frames in flight per rank (threads), gathers issued in step order by the main thread: every rank's k-th
    collective carries the same step, so frames never mix across steps."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, 29633, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_pipelined_frames_propagate_errors():
    from pcrender import multiview

    def render(i, slot):
        if i == 5:
            raise RuntimeError("boom")
        return torch.zeros(1)

    with pytest.raises(RuntimeError, match="boom"):
        multiview.run_frames_pipelined(render, 0, 12, 3, on_frame=lambda i, t: None, device=None)
    seen = []
    multiview.run_frames_pipelined(lambda i, s: torch.full((1,), float(i)), 0, 7, 3, on_frame=lambda i, t: seen.append(int(t)))
    assert seen == list(range(7))
