#!/usr/bin/env bash
# two-phase forward: parity (default cap and cap 1), per-view bench with several caps
set -u
OUT=$PWD/gpurun_out/r03d
mkdir -p $OUT
export TMPDIR=/tmp
for sc in random_aniso deep_stack capsule_circle big_splats opaque_early_stop; do
GSR_COOP_MAX_VIEWS=0 python scripts/debug/coop_diff.py dump $sc /tmp/a.npz && GSR_COOP_CAP=1 python scripts/debug/coop_diff.py dump $sc /tmp/b.npz && python scripts/debug/coop_diff.py cmp /tmp/a.npz /tmp/b.npz | grep "bad pixels"
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_batch.py tests/test_gpu_hostpath.py tests/test_gpu_configs.py -x -q -m gpu > $OUT/tests_default.log 2>&1
tail -2 $OUT/tests_default.log
GSR_COOP_CAP=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_api.py -x -q -m gpu > $OUT/tests_cap1.log 2>&1
tail -2 $OUT/tests_cap1.log
B="python $PWD/bench.py --steps 24 --warmup 12 --repeats 1 --no-cpu-baseline --no-per-view --streams 1 --views-per-call 1"
for cap in 32 64 128 256 512; do
(cd /tmp && GSR_COOP_CAP=$cap rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$cap -o t -- $B > $OUT/t$cap.log 2>&1)
echo cap $cap
python - $OUT/t$cap <<'PY'
import csv,glob,sys,os
f=glob.glob(os.path.join(sys.argv[1],"**","*kernel_stats.csv"),recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "render_forward" in n: print("  ", n[:50], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
done
find $OUT -name "*.csv" -size +2M -delete
