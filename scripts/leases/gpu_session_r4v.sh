#!/usr/bin/env bash
# round 4, session V (re-entry after the container was re-created): the driver's GPU suite + smoke + the default bench line
set -u
OUT=$PWD/gpurun_out/r4v
mkdir -p $OUT
export TMPDIR=/tmp
bash scripts/gpu_tests.sh r4v
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.json
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python - <<'PY'
import json
for f in ("bench_default", "bench_driver"):
    try:
        d = json.loads(open("gpurun_out/r4v/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("wall_over_gpu"), d.get("per_view_api_frames_per_s"), d["kernels_ms_per_frame"])
    except Exception as e:
        print(f, "unreadable", e)
PY
