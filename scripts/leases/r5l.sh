#!/usr/bin/env bash
# round 5, lease L: the backward walking COMPACTED consumed list prefixes (GSR_BWD_COMPACT=1, default here) against the lists themselves
set -u
OUT=$PWD/gpurun_out/r5l
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x -k "parity or fullsize or batch or api or configs or hostpath or fuzz" 2>&1 | grep -v amdgpu.ids | tail -3
for c in 1 0 1 0; do
  GSR_BWD_COMPACT=$c python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --drop-in-processes 0 > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err
  python - $OUT/bench_c$c.json $c <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels_ms_per_frame"]; di=d["drop_in_api"]
    print("compact=%s value %.1f  bwd %.4f items %.4f (in-region %.4f ms/launch) | 1-view bwd %.4f items %.4f in-order %.1f" % (sys.argv[2], d["value"], k["render_backward"], k["bwd_items"], d["roofline"]["avg_ms"], di["kernels_ms_per_frame"]["render_backward"], di["kernels_ms_per_frame"]["bwd_items"], di["frames_per_s"]["one_stream_in_order"]))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
done
