#!/usr/bin/env bash
# round 4, session D: full GPU suite on the new build, then the 16 000-case sweep three times (log kept under profiles/)
set -u
OUT=$PWD/gpurun_out/r4d
mkdir -p $OUT
python gaussian-pcloud-render_amd/build.py > /dev/null 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 3000 python scripts/fuzz_sweep.py 16000 --workers 16 --runs 3 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_16000_x3.log | tail -8
timeout 600 python scripts/bwd_accuracy.py 14139 14397 --repeat 3 2>&1 | grep -v amdgpu.ids | tee $OUT/acc_14139_14397.txt
