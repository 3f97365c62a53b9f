#!/usr/bin/env bash
# round 5, lease W: render backward grid (workgroup quartets per launch) for single-view submissions
set -u
OUT=$PWD/gpurun_out/r5w
mkdir -p $OUT
export TMPDIR=/tmp
summ() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
di=d["drop_in_api"]
print("%s: value %.1f | fwd %.4f bwd %.4f | 1-view fwd %.4f bwd %.4f sum %.4f in-order %.1f literal %.1f" % (sys.argv[2], d["value"], d["kernels_ms_per_frame"]["render_forward"], d["kernels_ms_per_frame"]["render_backward"], di["kernels_ms_per_frame"]["render_forward"], di["kernels_ms_per_frame"]["render_backward"], di["kernel_sum_ms_per_frame"], di["frames_per_s"]["one_stream_in_order"], di["frames_per_s"]["literal"]))
PY
}
for F in 4096 640 320 160 960; do
  GSR_BWD_FILL=$F python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --drop-in-processes 0 > $OUT/bench_f$F.json 2> $OUT/bench_f$F.err
  summ $OUT/bench_f$F.json "fill $F"
done
