#!/usr/bin/env bash
# round 5, lease AB: the 16 000-case randomised parity sweep x 2 on the final kernels (work-estimate launch order, dynamic backward hand-out)
set -u
OUT=$PWD/gpurun_out/r5ab
mkdir -p $OUT
export TMPDIR=/tmp
timeout 3000 python scripts/fuzz_sweep.py 16000 --workers 16 --runs 2 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_16000_x2.log | grep "^{\|FAILED" | tail -12
