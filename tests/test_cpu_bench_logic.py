"""CPU tests of bench.py's host logic (benchlib/): which camera view a rank renders at a step, the per-call-shape byte model against
SURVEY.md 8(d)'s formulas, the union of event intervals behind `gpu_ms_per_step_timed`, the roofline object's arithmetic and its
refusal to quote counters of another run, the CPU quota the baseline's thread count follows."""
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


def test_byte_model_equals_survey_8d_on_a_toy_scene():
    """SURVEY.md 8(d), per view: render forward 40 C + 8 T + 20 N, render backward 40 C' + 20 N + 44 V, pair emission
    20 P + (key + 4) R, tile sort passes x (key + 2 (key + 4)) R with 2-byte keys up to 65 536 tiles and 4-byte keys beyond."""
    sys.path.insert(0, ROOT)
    from benchlib.workload import algorithmic_bytes
    P, V, R, W, H, K, C, Cb = 1000, 900, 5000, 64, 48, 4, 700, 650
    T, N = 4 * 3, W * H
    b = algorithmic_bytes(P, V, R, T, N, K, C, Cb, tile_passes=1, views_per_call=1)
    assert b["render_forward"] == 40 * C + 8 * T + 20 * N
    assert b["render_backward"] == 40 * Cb + 20 * N + 44 * V
    assert b["duplicate"] == 20 * P + 6 * R and b["tile_sort"] == (2 + 2 * 6) * R
    assert b["preprocess"] == (44 + 12 * K) * P + 75 * V + 8 * (P - V)
    big = algorithmic_bytes(P, V, R, 70000, N, K, C, Cb, tile_passes=3, views_per_call=1)
    assert big["duplicate"] == 20 * P + 8 * R and big["tile_sort"] == 3 * (4 + 2 * 8) * R


def test_roofline_object_and_counter_provenance(tmp_path):
    sys.path.insert(0, ROOT)
    import argparse
    import json
    from benchlib import roofline as rf
    from benchlib.workload import kernels_sha
    args = argparse.Namespace(workload="toy", profile="training", forward_only=False)
    avg = {"render_backward": 2.0, "render_forward": 1.0, "preprocess": 0.3}
    bytes_per = {"render_backward": 1.0e8, "render_forward": 0.8e8, "preprocess": 1.0e8}
    key = {"workload": "toy", "points": 10, "width": 64, "height": 48, "views_per_launch": 12, "profile": "training", "forward_only": False}
    pmc = {"key": key, "kernels_sha": kernels_sha(), "lease": "test", "sclk_mhz": 2000.0,
           "bytes_per_launch": {"k_render_backward<2>": 2.5e9}, "avg_us": {"k_render_backward<2>": 2100.0, "k_render_forward<0>": 1000.0},
           "valu_wave_instructions_per_launch": {"k_render_backward<2>": 1.0e9},
           "valu_busy": {"k_render_backward<2>": {"valu_busy": 0.8}}, "valu_busy_formula": "f"}
    path = tmp_path / "pmc.json"
    path.write_text(json.dumps(pmc))
    r = rf.build(avg, {"render_backward": 2.2}, {"render_backward": [2.2] * 3}, bytes_per, 12, args, 10, 64, 48, [2100.0], [2100.0], pmc_path=str(path))
    # the dominant stage, its in-region duration, 12 views per launch: 1e8 x 12 bytes / 2.2 ms
    assert r["kernel"] == "render_backward" and r["avg_ms"] == 2.2
    assert abs(r["achieved"] - 1.0e8 * 12 / 2.2e-3 / 1e9) < 0.01 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4
    assert r["traffic"] == 2.5e9 and r["binding_frac"] == 0.8 and abs(r["issue_frac"] - 1.0e9 / 2.2e-3 / 1e9 / rf.VALU_PEAK_GWIPS) < 1e-3
    assert r["live_vs_profile"]["agree_within_10pct"]
    # counters of other kernel sources, or of another call shape, are not quoted
    for change in (dict(kernels_sha="0" * 16), dict(key=dict(key, views_per_launch=1))):
        path.write_text(json.dumps(dict(pmc, **change)))
        r2 = rf.build(avg, {"render_backward": 2.2}, {"render_backward": [2.2]}, bytes_per, 12, args, 10, 64, 48, [], [], pmc_path=str(path))
        assert r2["traffic"] is None and r2["binding_frac"] is None and r2["traffic_source"].startswith("null")
    assert rf.stage_kernel("render_forward", {"k_render_forward<0>": 1.0, "k_render_forward_half": 0.2}) == "k_render_forward<0>"
    assert rf.stage_kernel("render_backward", {"k_render_backward<2>": 1.0}) == "k_render_backward<2>"
    assert rf.build({}, {}, {}, {}, 1, args, 1, 1, 1, [], []) is None


def test_cpu_quota_is_read_from_the_cgroup(tmp_path):
    sys.path.insert(0, ROOT)
    from benchlib.cpu_baseline import cpu_quota_cores
    (tmp_path / "cpu.max").write_text("1600000 100000\n")
    assert cpu_quota_cores(str(tmp_path)) == 16.0
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert cpu_quota_cores(str(tmp_path)) is None
    assert cpu_quota_cores(str(tmp_path / "absent")) is None
