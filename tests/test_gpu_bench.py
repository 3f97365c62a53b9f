"""bench.py end to end on the GPU box: the single-rank contract line, and the multi-rank control flow (two ranks sharing
the one GPU through the gloo mode, since RCCL refuses two ranks per device) -- every rank has to issue the same
collectives in the same order or the run hangs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--workload", "synth-THuman-256", "--points", "20000", "--width", "256", "--height", "256", "--steps", "8", "--warmup", "2"]


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_single_rank_line_has_the_contract_fields(gpu_device):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--drop-in-processes", "2"] + SMALL, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["value"] > 0 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["repeats"] == 5 and len(d["ms_per_step_blocks"]) == 5 and "one_core" in d["cpu_baseline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    # the line explains its own wall time: GPU time of the timed steps from events inside the timed region, the shader clock
    # under load in the timed region and in the per-stage pass, and the untimed warm-up it really did
    assert d["gpu_ms_per_step_timed"] > 0 and len(d["gpu_ms_per_step_timed_blocks"]) == 5
    assert 0.9 < d["wall_over_gpu"] < 50
    assert d["gpu_ms_per_step_timed"] <= d["ms_per_step"] * 1.02
    assert 500 < d["sclk_mhz"]["timed_region"] < 3000 and 500 < d["sclk_mhz"]["stage_pass"] < 3000
    assert d["warmup_effective"]["seconds"] >= 1.0 and d["warmup_effective"]["steps"] >= d["warmup"]
    assert d["config"]["baseline_config"].startswith("configs[2]")
    # the co-headline: the reference's API as its caller uses it (settings built per call inside the timed loop, one stream, in
    # order), also from fresh processes; the roofline names the roof its numbers are quoted against AND the one that binds
    di = d["drop_in"]
    assert di["frames_per_s"] > 0 and 0 < di["fraction_of_value"] < 1.5 and "simple_raw_render.py:260-278" in di["what"]
    fp = di["fresh_processes"]
    assert fp["n"] == 2 and fp["errors"] is None and fp["min"] <= fp["median"] <= fp["max"] and fp["min"] > 0
    assert d["drop_in_api"]["frames_per_s"]["literal"] == di["frames_per_s"]
    assert set(("one_stream_in_order", "one_stream_overlap_opt_in", "four_streams")) <= set(d["drop_in_api"]["frames_per_s"])
    assert d["overlap_of_consecutive_calls"]["on"] is False
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["frac"] == rf["hbm_frac"] and "binding_roof" in rf and len(d["kernels_sha"]) == 16
    assert rf["traffic"] is None or d["kernels_sha"] in rf["traffic_source"]
    assert "no run on more than one GPU has been measured" in d["multi_gpu_status"]


def test_two_ranks_complete(gpu_device):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--device-index", "0",
           "--no-cpu-baseline"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["cpu_baseline"] is None


def test_rccl_backend_runs_at_world_1(gpu_device):
    """The `nccl` (RCCL) backend itself, which the gloo stand-in above never touches: one rank, device tensors, the sharded
    render + frame gather + gradient all-reduce of pcrender.multiview."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_worker.py"), "29543"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_config3_full_size_frames_through_rccl_at_world_1(gpu_device):
    """BASELINE configs[3] at full size as far as one GPU goes: 8 x 800K / 1080p frames through pcrender.multiview's gather on the
    RCCL backend with full-size buffers, both gather modes, frames equal to per-view calls (tests/rccl_config3_worker.py)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_config3_worker.py"), "29544"], capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_CONFIG3_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_bench_under_torchrun_one_rank_uses_rccl(gpu_device):
    """bench.py launched the way the driver launches it for N > 1, with N = 1: process group on `nccl`, frames gathered
    through RCCL, timing reduced with an RCCL all-reduce."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29542", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline", "--repeats", "2"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0 and "RCCL frame gather" in d["config"]["workload"] and d["repeats"] == 2


def test_gpus_flag_launches_ranks_or_fails_loudly(gpu_device):
    """`python bench.py --gpus 2` without torch.distributed.run: runs two RCCL ranks when the box has two GPUs, otherwise
    refuses with a message (it must never quietly run one rank and print n_gpus: 1)."""
    import torch
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline", "--repeats", "1"] + SMALL,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    if torch.cuda.device_count() >= 2:
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        assert _last_json(r.stdout)["n_gpus"] == 2
    else:
        assert r.returncode != 0 and "only 1 GPU(s) visible" in (r.stdout + r.stderr)
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_gpus_flag_self_launch_two_ranks_on_one_gpu_via_gloo(gpu_device):
    """The self-launch path end to end on a one-GPU box: --gpus 2 with the gloo stand-in and both ranks on device 0."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--device-index", "0",
                        "--no-cpu-baseline", "--repeats", "1"] + SMALL, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert _last_json(r.stdout)["n_gpus"] == 2


def test_config3_shape_two_ranks_p2p_gather_on_one_gpu(gpu_device):
    """BASELINE configs[3]'s shape at a small size: 8 views dealt round-robin to the ranks (4 per rank at world 2, one call per
    step), frames gathered on rank 0 with the grouped send / receive fallback; the line carries per-rank rates and the exposed
    gather time."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--device-index", "0",
                        "--no-cpu-baseline", "--repeats", "2", "--config", "3", "--gather-mode", "p2p", "--no-per-view",
                        "--workload", "synth-THuman-256", "--points", "20000", "--width", "256", "--height", "256", "--steps", "8",
                        "--warmup", "2"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["call_shape"]["views_per_call"] == 4
    assert d["config"]["baseline_config"].startswith("configs[3]")
    assert len(d["per_rank_frames_per_s"]) == 2 and min(d["per_rank_frames_per_s"]) > 0
    ga = d["distributed"]["gather"]
    assert ga["mode"] == "p2p" and len(ga["exposed_ms_per_block"]) == 2


@pytest.mark.parametrize("config", [3, 4])
def test_eight_rank_rehearsal_on_one_gpu(config, gpu_device):
    """BASELINE configs[3] / [4] as the round driver would launch them on an 8-GPU node (`bench.py --gpus 8 --config N`: 8 views, one
    per rank, frames gathered on rank 0), rehearsed with eight gloo ranks sharing this box's one GPU at a tiny size: every rank
    issues the same collectives in the same order (or the run hangs), the line carries one rate per rank, where each rank ran and
    its capped host thread count, and says that no multi-GPU number has been measured."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dist-backend", "gloo", "--device-index", "0",
                        "--no-cpu-baseline", "--repeats", "1", "--config", str(config), "--no-per-view", "--warmup-seconds", "0.2",
                        "--workload", "synth-THuman-256", "--points", "6000", "--width", "160", "--height", "96", "--steps", "4",
                        "--warmup", "1"], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 8 and d["config"]["baseline_config"].startswith("configs[%d]" % config)
    assert d["config"]["call_shape"]["views_per_call"] == 1          # 8 views over 8 ranks: one view per rank and submission
    assert len(d["per_rank_frames_per_s"]) == 8 and min(d["per_rank_frames_per_s"]) > 0
    dd = d["distributed"]
    assert dd["world"] == 8 and [x["rank"] for x in dd["ranks"]] == list(range(8))
    assert [x["local_rank"] for x in dd["ranks"]] == list(range(8)) and all(x["device"] == 0 for x in dd["ranks"])
    cores = os.cpu_count() or 1
    assert all(1 <= x["host_threads"] <= max(1, cores // 8) for x in dd["ranks"]), dd["ranks"]
    assert dd["gather"]["collectives"] >= 4 and d["cpu_baseline"] is None
    assert "no run on more than one GPU has been measured" in d["multi_gpu_status"]
