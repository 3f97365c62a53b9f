#!/usr/bin/env bash
# round 6, lease N: where the emission workgroups' time goes (instrumentation build), tickets on / off
set -u
OUT=$PWD/gpurun_out/${LEASE:-r6n}
mkdir -p $OUT
export TMPDIR=/tmp
cp gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so /tmp/libgsr_plain.so
GSR_EXTRA_FLAGS="-DGSR_STATS" python gaussian-pcloud-render_amd/build.py --force > $OUT/build.log 2>&1; tail -1 $OUT/build.log
for tk in 1 0; do echo "== GSR_TICKETS=$tk"; GSR_TICKETS=$tk python scripts/dup_times.py 2>&1 | grep -v amdgpu.ids | tail -6; done
cp /tmp/libgsr_plain.so gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so
