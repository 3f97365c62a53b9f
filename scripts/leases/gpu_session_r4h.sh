#!/usr/bin/env bash
set -u
OUT=$PWD/gpurun_out/r4h
mkdir -p $OUT
python gaussian-pcloud-render_amd/build.py > /dev/null 2>&1
timeout 600 python scripts/probe/scatter_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/scatter_probe.txt
timeout 600 python scripts/probe/scatter_probe.py --chunk 256 2>&1 | grep -v amdgpu.ids | tee -a $OUT/scatter_probe.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-per-view > $OUT/d12.json 2> $OUT/d12.err
python - $OUT/d12.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], "fps ms/step", d["ms_per_step"], "blocks", d["ms_per_step_blocks"], "subs", d["gpu_ms_submissions_blocks"], "wall/gpu", d["wall_over_gpu"])
PY
