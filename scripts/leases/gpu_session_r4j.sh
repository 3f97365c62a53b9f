#!/usr/bin/env bash
# round 4, session J: overlap of consecutive per-view calls inside the library; full suite; bench with the drop-in figures
set -u
OUT=$PWD/gpurun_out/r4j
mkdir -p $OUT
python gaussian-pcloud-render_amd/build.py > /dev/null 2>&1
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/bench_default.err
python - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], "fps ms/step", d["ms_per_step"], "blocks", d["ms_per_step_blocks"], "wall/gpu", d["wall_over_gpu"], "sclk", d["sclk_mhz"]["timed_blocks"], d["sclk_mhz"]["stage_pass"])
print("kernels/frame", d["kernels_ms_per_frame"], "sum", round(sum(d["kernels_ms_per_frame"].values()),4))
print("drop_in", d["drop_in_api"]["frames_per_s"], d["drop_in_api"].get("overlapped_calls"), "fwd_only", d["forward_only"], "rgb", d["rgb_time_equiv"].get("ms_per_12_views"))
PY
