#!/usr/bin/env bash
# round 5, lease R: half-quadrant forward with 128-entry rounds: tests and kernel times
set -u
OUT=$PWD/gpurun_out/r5r
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x -k "parity or fullsize or batch or api or configs or hostpath or footprint or passes or fuzz" 2>&1 | grep -v amdgpu.ids | tail -3
python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --drop-in-processes 0 > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
di=d["drop_in_api"]
print("value %.1f | fwd %.4f bwd %.4f | 1-view fwd %.4f sum %.4f in-order %.1f | fwd-only %s" % (d["value"], d["kernels_ms_per_frame"]["render_forward"], d["kernels_ms_per_frame"]["render_backward"], di["kernels_ms_per_frame"]["render_forward"], di["kernel_sum_ms_per_frame"], di["frames_per_s"]["one_stream_in_order"], d["forward_only"]))
PY
