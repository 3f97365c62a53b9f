"""CPU tests of bench.py's host logic: which camera view a rank renders at a step, the per-call-shape byte model, the
union of event intervals behind `gpu_ms_per_step_timed`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_view_assignment_gives_distinct_views_for_every_world_size():
    """`(i * world + rank) % 12` handed a rank 3 distinct views x 4 copies at world 8.  Now any 12
    consecutive steps of a rank are the 12 views, for every world size, ranks start at different views, and the sharded
    configs deal each view of a turn to exactly one rank."""
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 3, 4, 5, 8, 12):
        for rank in range(world):
            for first in (0, 5, 12, 31):
                assert sorted(bench.view_of(first + k, rank, world, 12, "circle") for k in range(12)) == list(range(12))
        if world <= 12:
            for k in (0, 7):
                assert len({bench.view_of(k, r, world, 12, "circle") for r in range(world)}) == world
    for world in (1, 2, 4, 8):
        owners = {}
        for rank in range(world):
            per = 8 // world
            for k in range(per):
                owners.setdefault(bench.view_of(k, rank, world, 8, "views"), []).append(rank)
        assert sorted(owners) == list(range(8)) and all(len(v) == 1 for v in owners.values())


def test_byte_model_counts_shared_traffic_once_per_batch():
    sys.path.insert(0, ROOT)
    import bench
    kw = dict(P=800_000, V=800_000, R=11_800_000, T=8160, N=1920 * 1080, K=4, C_fwd=1.9e6, C_bwd=1.9e6, tile_passes=2)
    one, twelve = bench.algorithmic_bytes(views_per_call=1, **kw), bench.algorithmic_bytes(views_per_call=12, **kw)
    assert "offsets_scan" not in one                                   # the prefix sum is part of the pair emission
    # SURVEY 8(d)'s per-view figures at one view per call ...
    assert one["preprocess"] == (44 + 48) * 800_000 + 75 * 800_000
    assert one["preprocess_backward"] == (92 + 107 + 48 + 40 + 48) * 800_000
    # ... and the cloud read / the gradients written once per 12-view batch
    assert twelve["preprocess"] == (44 + 48) * 800_000 / 12 + 75 * 800_000
    assert abs(twelve["preprocess_backward"] - (92 + (107 + 48 + 40 + 48) / 12) * 800_000) < 1
    for k in ("depth_sort", "duplicate", "tile_sort", "render_forward", "render_backward"):
        assert one[k] == twelve[k]


def test_union_of_event_intervals():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.union_ms([]) == 0.0
    assert bench.union_ms([(0.0, 2.0), (1.0, 3.0), (5.0, 6.0)]) == 4.0
    assert bench.union_ms([(5.0, 6.0), (0.0, 10.0)]) == 10.0
