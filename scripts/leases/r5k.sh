#!/usr/bin/env bash
# round 5, lease K: what depth hints are worth on the other configurations (bench side figure `depth_hint`), after the hint tests
set -u
OUT=$PWD/gpurun_out/r5k
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hints.py tests/test_gpu_parity.py -q -m gpu -k "hint or half_quadrant or subquadrant" 2>&1 | grep -v amdgpu.ids | tail -3
for cfg in 1 4 3; do
  python bench.py --config $cfg --steps 16 --warmup 8 --repeats 2 --no-cpu-baseline --drop-in-processes 0 > $OUT/bench_c$cfg.json 2> $OUT/bench_c$cfg.err
  python - $OUT/bench_c$cfg.json $cfg <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h=d["depth_hint"]
    print("config %s value %.1f | hints %s vs %s same loop, repeats %s | kernels with hints %s | without %s" % (sys.argv[2], d["value"], h.get("frames_per_s"), h.get("frames_per_s_without_hints_same_loop"), h.get("repeated"), h.get("kernels_ms_per_frame"), d["kernels_ms_per_frame"]))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
done
