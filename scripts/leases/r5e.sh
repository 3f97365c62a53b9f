#!/usr/bin/env bash
# round 5, lease E: the half-quadrant forward for single-view submissions (GSR_FWD_HALF_V=1, the default) against the 8 x 8 kernel
# (GSR_FWD_HALF_V=0): the -m gpu suite in both modes, then the bench's per-view figures of each
set -u
OUT=$PWD/gpurun_out/r5e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x -k "parity or fullsize or footprint or adversarial or hostpath or batch or api" 2>&1 | grep -v amdgpu.ids | tail -3
grep -E "^FAILED|^ERROR" $OUT/tests.log | head -20
GSR_FWD_HALF_V=0 timeout 1500 python -m pytest tests -q -m gpu -x -k "parity or fullsize or footprint or adversarial or hostpath" 2>&1 | grep -v amdgpu.ids | tail -3
for hv in 1 0 1 0 1 0; do
  GSR_FWD_HALF_V=$hv python bench.py --steps 24 --warmup 12 --repeats 2 --no-cpu-baseline --drop-in-processes 0 > $OUT/bench_half$hv.json 2> $OUT/bench_half$hv.err
  python - $OUT/bench_half$hv.json $hv <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    di=d["drop_in_api"]
    print("half_v=%s value %.1f | per-view %s | fwd-only %s | 1-view kernels fwd %.4f bwd %.4f sum %.4f" % (sys.argv[2], d["value"], di["frames_per_s"], d["forward_only"], di["kernels_ms_per_frame"]["render_forward"], di["kernels_ms_per_frame"]["render_backward"], di["kernel_sum_ms_per_frame"]))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
done
