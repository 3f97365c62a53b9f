#!/usr/bin/env bash
set -u
OUT=$PWD/gpurun_out/r03c
mkdir -p $OUT
export GSR_COOP_MAX_VIEWS=2
GSR_EXTRA_FLAGS="-DGSR_STATS" python gaussian-pcloud-render_amd/build.py --force > $OUT/build_stats.log 2>&1
tail -3 $OUT/build_stats.log
python scripts/coop_times.py 2>&1 | tee $OUT/coop_times.txt
GSR_ORDER_FOLD=4096 python scripts/coop_times.py 2>&1 | tee $OUT/coop_times_fold.txt
python gaussian-pcloud-render_amd/build.py --force > /dev/null 2>&1
