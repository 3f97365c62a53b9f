#!/usr/bin/env bash
# round 6, lease O: occupancy of the pair emission (k_duplicate is latency-bound per workgroup: 16 us gather + 19 us emission at 5 waves per SIMD)
set -u
OUT=$PWD/gpurun_out/${LEASE:-r6o}
mkdir -p $OUT
export TMPDIR=/tmp
cp gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so /tmp/libgsr_plain.so
for v in "-DGSR_DUP_WAVES=5" "-DGSR_DUP_WAVES=6" "-DGSR_DUP_WAVES=7" "-DGSR_DUP_WAVES=8 -DGSR_DUP_G=2" "-DGSR_DUP_WAVES=6 -DGSR_DUP_G=2" "-DGSR_DUP_WAVES=5 -DGSR_DUP_G=8"; do
GSR_EXTRA_FLAGS="$v" python gaussian-pcloud-render_amd/build.py --force > $OUT/build.log 2>&1
for vpc in 12 1; do
python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --no-per-view --views-per-call $vpc > $OUT/b.json 2>$OUT/b.err
python - $OUT/b.json "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels_ms_per_frame"]
    print("%-40s %2d views/call: value %7.1f duplicate %.4f sum %.4f" % (sys.argv[2], d["views_per_call"], d["value"], k["duplicate"], sum(k.values())))
except Exception as e:
    print("no result", sys.argv[2], e); print(open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
done
done
cp /tmp/libgsr_plain.so gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so
