#!/usr/bin/env bash
# round 5, lease Z: batch forward pulling work units (no 290 k empty workgroups at the end of the launch): tests, kernel times, timeline
set -u
OUT=$PWD/gpurun_out/r5z
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x -k "parity or fullsize or batch or api or configs or hostpath or footprint or passes or fuzz" 2>&1 | grep -v amdgpu.ids | tail -3
for i in 1 2; do
python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --drop-in-processes 0 > $OUT/bench$i.json 2> $OUT/bench$i.err
python - $OUT/bench$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
di=d["drop_in_api"]
print("value %.1f | fwd %.4f bwd %.4f sum %.4f | 1-view fwd %.4f bwd %.4f sum %.4f in-order %.1f literal %.1f | fwd-only %s rgb %s" % (d["value"], d["kernels_ms_per_frame"]["render_forward"], d["kernels_ms_per_frame"]["render_backward"], sum(d["kernels_ms_per_frame"].values()), di["kernels_ms_per_frame"]["render_forward"], di["kernels_ms_per_frame"]["render_backward"], di["kernel_sum_ms_per_frame"], di["frames_per_s"]["one_stream_in_order"], di["frames_per_s"]["literal"], d["forward_only"], d["rgb_time_equiv"]["ms_per_12_views"]))
PY
done
cp gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so /tmp/libgsr_plain.so
GSR_EXTRA_FLAGS="-DGSR_STATS" python gaussian-pcloud-render_amd/build.py --force > $OUT/build.log 2>&1; tail -1 $OUT/build.log
python scripts/debug/fwd_half_tail.py 0 12 > $OUT/tail_12.txt 2> $OUT/err.txt; head -18 $OUT/tail_12.txt; tail -3 $OUT/err.txt
cp /tmp/libgsr_plain.so gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so
