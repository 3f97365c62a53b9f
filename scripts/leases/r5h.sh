#!/usr/bin/env bash
# round 5, lease H: k_tile_order / k_bwd_items with their per-thread loads hoisted into registers; the sub-quadrant moments test
set -u
OUT=$PWD/gpurun_out/r5h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x -k "parity or fullsize or batch or api or configs or hostpath or footprint" 2>&1 | grep -v amdgpu.ids | tail -4
python bench.py --steps 24 --warmup 12 --repeats 2 --no-cpu-baseline --drop-in-processes 0 > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
di=d["drop_in_api"]
print("value %.1f | per-view %s | 1-view kernels %s sum %.4f | 12-view %s" % (d["value"], di["frames_per_s"], di["kernels_ms_per_frame"], di["kernel_sum_ms_per_frame"], d["kernels_ms"]))
PY
