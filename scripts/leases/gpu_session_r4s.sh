#!/usr/bin/env bash
# round 4, session S: final build: suite, the round-4 profile set again (one lease), the driver's shape
set -u
python gaussian-pcloud-render_amd/build.py > /dev/null 2>&1
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash scripts/profile_gpu.sh r04 2>&1 | tail -5
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/prof_r04/bench_driver_shape.json 2> gpurun_out/prof_r04/bench_driver_shape.err
python - <<'PY'
import json
for f in ("gpurun_out/prof_r04/bench.json","gpurun_out/prof_r04/bench_driver_shape.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d["kernels_ms_per_frame"]
    print(f.split("/")[-1], d["value"], "fps blocks", d["ms_per_step_blocks"], "wall/gpu", d["wall_over_gpu"], "sclk", d["sclk_mhz"]["timed_blocks"], "| roofline avg_ms", d["roofline"]["avg_ms"], d["roofline"]["avg_ms_stage_pass"], d["roofline"]["frac"], d["roofline"].get("live_vs_profile"), "| drop_in", d["drop_in_api"]["frames_per_s"])
PY
