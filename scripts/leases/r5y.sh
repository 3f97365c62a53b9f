#!/usr/bin/env bash
# round 5, lease Y: instrumented build -- timeline of the 12-view forward launch (is the dispatch of the empty tiles' workgroups exposed?)
set -u
OUT=$PWD/gpurun_out/r5y
mkdir -p $OUT
export TMPDIR=/tmp
cp gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so /tmp/libgsr_plain.so
GSR_EXTRA_FLAGS="-DGSR_STATS" python gaussian-pcloud-render_amd/build.py --force > $OUT/build.log 2>&1; tail -1 $OUT/build.log
python scripts/debug/fwd_half_tail.py 0 12 > $OUT/tail_12.txt 2> $OUT/err.txt; head -40 $OUT/tail_12.txt; tail -3 $OUT/err.txt
cp /tmp/libgsr_plain.so gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so
