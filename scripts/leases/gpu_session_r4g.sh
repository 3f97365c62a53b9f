#!/usr/bin/env bash
# round 4, session G: new config tests; the driver's bench shape with 12+8 and with 10+10 views per call
set -u
OUT=$PWD/gpurun_out/r4g
mkdir -p $OUT
python gaussian-pcloud-render_amd/build.py > /dev/null 2>&1
timeout 2400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_bench.py tests/test_gpu_api.py tests/test_gpu_hostpath.py -x -q -m gpu 2>&1 | tail -6
for v in "d12|" "d10|--views-per-call 10"; do
  name=${v%%|*}; flags=${v#*|}
  timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-per-view $flags > $OUT/$name.json 2> $OUT/$name.err; tail -c 300 $OUT/$name.err
done
python - $OUT/d12.json $OUT/d10.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d["value"], "fps ms/step", d["ms_per_step"], "blocks", d["ms_per_step_blocks"], "gpu", d["gpu_ms_per_step_timed_blocks"], "subs", d["gpu_ms_submissions_blocks"], "wall/gpu", d["wall_over_gpu"], "sclk", d["sclk_mhz"]["timed_region"], d["sclk_mhz"]["stage_pass"])
        print("   kernels/frame", d["kernels_ms_per_frame"], "sum", round(sum(d["kernels_ms_per_frame"].values()),4))
    except Exception as e:
        print(f, "no result", e)
PY
