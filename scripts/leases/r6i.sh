#!/usr/bin/env bash
# round 6, lease I: threshold of the adaptive sub-quadrant switch: (b / sigma)^2 > 16 instead of 20 (A/B build)
set -u
OUT=$PWD/gpurun_out/${LEASE:-r6i}
mkdir -p $OUT
export TMPDIR=/tmp
cp gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so /tmp/libgsr_plain.so
GSR_EXTRA_FLAGS="-DGSR_SUBQ_M=16.f" python gaussian-pcloud-render_amd/build.py --force > $OUT/build.log 2>&1; tail -1 $OUT/build.log
GSR_BWD_SUBQ=2 python scripts/bwd_accuracy.py 149 14139 14397 --range 0 500 > $OUT/acc_2.txt 2> $OUT/acc_2.err; tail -5 $OUT/acc_2.txt
for rep in 1 2; do
for mode in 0 2; do
for vpc in 12 1; do
GSR_BWD_SUBQ=$mode python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --no-per-view --views-per-call $vpc > $OUT/b_${mode}_${vpc}_$rep.json 2>$OUT/b_${mode}_${vpc}_$rep.err
python - $OUT/b_${mode}_${vpc}_$rep.json $mode <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels_ms_per_frame"]
    print("M=16 subq mode", sys.argv[2], d["views_per_call"], "views/call:", d["value"], "fps; render_backward %.4f (in-region %s) sum %.4f" % (k["render_backward"], d["roofline"].get("avg_ms"), sum(k.values())))
except Exception as e:
    print("no result", e); print(open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
done
done
cp /tmp/libgsr_plain.so gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so
