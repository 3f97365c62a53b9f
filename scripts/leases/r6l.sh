#!/usr/bin/env bash
# round 6, lease L: the 16 000-case randomised parity sweep x 2 against the reference build on this round's kernels (adaptive
# sub-quadrant moments default, blockIdx-numbered emission), and once more with the look-back sorts (GSR_SORT_MODE=1)
set -u
OUT=$PWD/gpurun_out/${LEASE:-r6l}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 3000 python scripts/fuzz_sweep.py 16000 --workers 16 --runs 2 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_16000_x2.log | grep "^{\|FAILED" | tail -12
GSR_SORT_MODE=1 timeout 1500 python scripts/fuzz_sweep.py 4000 --workers 16 --runs 1 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_4000_lookback.log | grep "^{\|FAILED" | tail -6
