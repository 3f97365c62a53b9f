#!/usr/bin/env bash
# round 6, lease K: bench.py after the split into benchlib/ (driver's command line), staging groups 44 words apart in the render backward
set -u
OUT=$PWD/gpurun_out/${LEASE:-r6k}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x -k "bench or parity or bwd_batches or batch" 2>&1 | grep -v amdgpu.ids | tail -2
(time python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err) 2>&1 | grep real
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "| roofline", {k: r.get(k) for k in ("kernel","frac","binding_frac","issue_frac","avg_ms","traffic")})
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value","cores","cpu_quota_cores","host_cpus")})
print("drop_in", d["drop_in"]["frames_per_s"], d["drop_in"]["host_exposed_us"], d["drop_in"]["fresh_processes"])
print("anchor", d.get("gather_world1_anchor"))
print("kernels", d["kernels_ms_per_frame"])
PY
tail -2 $OUT/bench.err
for i in 1 2; do
python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --no-per-view > $OUT/b$i.json 2>$OUT/b$i.err
python - $OUT/b$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "render_backward in-region", d["roofline"]["avg_ms"], d["kernels_ms_per_frame"]["render_backward"])
PY
done
