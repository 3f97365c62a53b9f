#!/usr/bin/env bash
# round 6, lease F: the per-view call's short host path (forward_view / backward_view, lazy nn.Module): tests, literal host trace, bench
set -u
OUT=$PWD/gpurun_out/${LEASE:-r6f}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x -k "api or hostpath or parity or batch or integration or passes or bwd_batches" 2>&1 | grep -v amdgpu.ids | tail -3
python scripts/literal_host_trace.py 240 > $OUT/trace.json 2> $OUT/trace.err; cat $OUT/trace.json; tail -2 $OUT/trace.err
python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --drop-in-processes 3 > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
di=d["drop_in_api"]
print("value %.1f | drop_in %s | exposed %s | api %s | 1-view kernels %s sum %.4f" % (d["value"], d["drop_in"]["frames_per_s"], di.get("host_exposed_us"), di["frames_per_s"], di["kernels_ms_per_frame"], di["kernel_sum_ms_per_frame"]))
print("fresh", d["drop_in"]["fresh_processes"])
PY
tail -3 $OUT/bench.err
