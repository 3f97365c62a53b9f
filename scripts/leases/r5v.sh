#!/usr/bin/env bash
# round 5, lease V: instrumented build -- the render backward's packing at 1 and 12 views per call
set -u
OUT=$PWD/gpurun_out/r5v
mkdir -p $OUT
export TMPDIR=/tmp
cp gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so /tmp/libgsr_plain.so
GSR_EXTRA_FLAGS="-DGSR_STATS" python gaussian-pcloud-render_amd/build.py --force > $OUT/build.log 2>&1; tail -1 $OUT/build.log
python scripts/debug/bwd_tail.py > $OUT/bwd_tail.txt 2> $OUT/err.txt; cat $OUT/bwd_tail.txt; tail -3 $OUT/err.txt
cp /tmp/libgsr_plain.so gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so
