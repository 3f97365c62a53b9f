#!/usr/bin/env bash
# round 4, session F: full suite, 16 000-case sweep x3, bench default + the driver's shape (--steps 20 --warmup 5)
set -u
OUT=$PWD/gpurun_out/r4f
mkdir -p $OUT
python gaussian-pcloud-render_amd/build.py > /dev/null 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 3000 python scripts/fuzz_sweep.py 16000 --workers 16 --runs 3 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_16000_x3.log | grep "^{\|FAILED"
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err; tail -c 600 $OUT/bench_driver_shape.err
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
python - $OUT/bench_driver_shape.json $OUT/bench_default.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d["value"], "fps ms/step", d["ms_per_step"], "blocks", d["ms_per_step_blocks"], "gpu_ms_per_step_timed", d["gpu_ms_per_step_timed"], "wall/gpu", d["wall_over_gpu"], "sclk", d["sclk_mhz"]["timed_region"], d["sclk_mhz"]["timed_region_min_max"], d["sclk_mhz"]["stage_pass"], "warm", d["warmup_effective"]["steps"], d["warmup_effective"]["seconds"])
        print("   kernels/frame", d["kernels_ms_per_frame"], "sum", round(sum(d["kernels_ms_per_frame"].values()),4), "frame_hbm", d["frame_hbm"], "drop_in", (d.get("drop_in_api") or {}).get("frames_per_s"))
    except Exception as e:
        print(f, "no result", e)
PY
