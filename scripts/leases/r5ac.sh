#!/usr/bin/env bash
# round 5, lease AC: single-view forward under the work-estimate order: half-quadrant kernel (8 waves per tile, 4 per SIMD) vs the 8 x 8 kernel (4 per tile, 5 per SIMD)
set -u
OUT=$PWD/gpurun_out/r5ac
mkdir -p $OUT
export TMPDIR=/tmp
run() { tag=$1; shift
  env "$@" python bench.py --steps 24 --warmup 12 --repeats 2 --no-cpu-baseline --drop-in-processes 0 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - $OUT/bench_$tag.json $tag <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
di=d["drop_in_api"]
print("%s: 1-view fwd %.4f bwd %.4f sum %.4f in-order %.1f literal %.1f | fwd-only per-view %.1f" % (sys.argv[2], di["kernels_ms_per_frame"]["render_forward"], di["kernels_ms_per_frame"]["render_backward"], di["kernel_sum_ms_per_frame"], di["frames_per_s"]["one_stream_in_order"], di["frames_per_s"]["literal"], d["forward_only"]["per_view_call_frames_per_s"]))
PY
}
run half A=1
run full GSR_FWD_HALF_V=0
run full_k0 GSR_FWD_HALF_V=0 GSR_ORDER_KNEE=0
run half2 A=1
run full2 GSR_FWD_HALF_V=0
