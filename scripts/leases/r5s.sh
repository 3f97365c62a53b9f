#!/usr/bin/env bash
# round 5, lease S: instrumented build (GSR_STATS) -- what the single-view forward's last waves do
set -u
OUT=$PWD/gpurun_out/r5s
mkdir -p $OUT
export TMPDIR=/tmp
cp gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so /tmp/libgsr_plain.so
GSR_EXTRA_FLAGS="-DGSR_STATS" python gaussian-pcloud-render_amd/build.py --force > $OUT/build.log 2>&1; tail -2 $OUT/build.log
python scripts/debug/fwd_half_tail.py 0 > $OUT/tail_v0.txt 2> $OUT/tail.err; cat $OUT/tail_v0.txt
python scripts/debug/fwd_half_tail.py 5 > $OUT/tail_v5.txt 2>> $OUT/tail.err; cat $OUT/tail_v5.txt
tail -3 $OUT/tail.err
cp /tmp/libgsr_plain.so gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so
