#!/usr/bin/env bash
# round 5, lease U: half-quadrant forward at five waves per SIMD (96 registers, 19 spilled) against four; order estimate exponents
set -u
OUT=$PWD/gpurun_out/r5u
mkdir -p $OUT
export TMPDIR=/tmp
summ() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
di=d["drop_in_api"]
print("%s: value %.1f | fwd %.4f bwd %.4f | 1-view fwd %.4f bwd %.4f sum %.4f in-order %.1f literal %.1f | fwd-only %s" % (sys.argv[2], d["value"], d["kernels_ms_per_frame"]["render_forward"], d["kernels_ms_per_frame"]["render_backward"], di["kernels_ms_per_frame"]["render_forward"], di["kernels_ms_per_frame"]["render_backward"], di["kernel_sum_ms_per_frame"], di["frames_per_s"]["one_stream_in_order"], di["frames_per_s"]["literal"], d["forward_only"]))
PY
}
run() { # tag, env...
  tag=$1; shift
  env "$@" python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --drop-in-processes 0 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  summ $OUT/bench_$tag.json "$tag"
}
run k0_w4 GSR_ORDER_KNEE=0
run e03_w4 GSR_ORDER_EXP=0.3
run e03_w5 GSR_ORDER_EXP=0.3 GSR_FWD_HALF_WAVES=5
run e06_w5 GSR_ORDER_EXP=0.6 GSR_FWD_HALF_WAVES=5
run k0_w5 GSR_ORDER_KNEE=0 GSR_FWD_HALF_WAVES=5
run e03_w4b GSR_ORDER_EXP=0.3
