#!/usr/bin/env bash
# round 5, lease M: what the round driver runs, on the final tree: the -m gpu suite, smoke(), python bench.py --steps 20 --warmup 5
set -u
OUT=$PWD/gpurun_out/r5m
mkdir -p $OUT
export TMPDIR=/tmp
PYTEST_X= bash scripts/gpu_tests.sh r5m
grep -E "^FAILED|^ERROR" $OUT/tests.log | head
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5m/bench_driver.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "drop_in", d["drop_in"]["frames_per_s"], d["drop_in"]["fresh_processes"]["frames_per_s"], "traffic", d["roofline"]["traffic"], d["roofline"]["binding_frac"], "cpu", d["cpu_baseline"]["value"])
PY
