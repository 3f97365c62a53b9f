#!/usr/bin/env bash
# round 5, lease J: depth hints through the public API, and what they are worth on the headline workload (bench side figure)
set -u
OUT=$PWD/gpurun_out/r5j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hints.py -x -q -m gpu -s 2>&1 | grep -v amdgpu.ids | tail -12
python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --drop-in-processes 0 > $OUT/bench.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f" % d["value"], "| depth_hint", json.dumps(d["depth_hint"]))
print("kernels ms/frame (no hints)", d["kernels_ms_per_frame"])
PY
