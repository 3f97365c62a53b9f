#!/usr/bin/env bash
# round 5, lease T: render launch order by a work estimate instead of the list length: instrumented timeline, then A/B of the plain build
set -u
OUT=$PWD/gpurun_out/r5t
mkdir -p $OUT
export TMPDIR=/tmp
summ() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
di=d["drop_in_api"]
print("%s: value %.1f | fwd %.4f bwd %.4f | 1-view fwd %.4f bwd %.4f sum %.4f in-order %.1f literal %.1f | fwd-only %s" % (sys.argv[2], d["value"], d["kernels_ms_per_frame"]["render_forward"], d["kernels_ms_per_frame"]["render_backward"], di["kernels_ms_per_frame"]["render_forward"], di["kernels_ms_per_frame"]["render_backward"], di["kernel_sum_ms_per_frame"], di["frames_per_s"]["one_stream_in_order"], di["frames_per_s"]["literal"], d["forward_only"]))
PY
}
for K in 0 2048 1536 3072; do
  GSR_ORDER_KNEE=$K python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --drop-in-processes 0 > $OUT/bench_k$K.json 2> $OUT/bench_k$K.err
  summ $OUT/bench_k$K.json "knee $K"
done
GSR_ORDER_KNEE=2048 GSR_ORDER_EXP=1.0 python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --drop-in-processes 0 > $OUT/bench_k2048e1.json 2> $OUT/bench_e.err
summ $OUT/bench_k2048e1.json "knee 2048 exp 1.0"
GSR_ORDER_KNEE=2048 GSR_ORDER_EXP=0.3 python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --drop-in-processes 0 > $OUT/bench_k2048e03.json 2> $OUT/bench_e.err
summ $OUT/bench_k2048e03.json "knee 2048 exp 0.3"
cp gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so /tmp/libgsr_plain.so
GSR_EXTRA_FLAGS="-DGSR_STATS" python gaussian-pcloud-render_amd/build.py --force > $OUT/build.log 2>&1; tail -1 $OUT/build.log
for K in 0 2048; do
  GSR_ORDER_KNEE=$K python scripts/debug/fwd_half_tail.py 0 > $OUT/tail_k$K.txt 2> $OUT/tail.err
  echo "== knee $K"; head -14 $OUT/tail_k$K.txt; grep -A9 "END last" $OUT/tail_k$K.txt
done
cp /tmp/libgsr_plain.so gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so
