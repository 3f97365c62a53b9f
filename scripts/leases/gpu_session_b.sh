#!/usr/bin/env bash
# kernel-change session: parity subset on the default build, then the 12-views-per-call kernel times of the variants in $1
set -u
OUT=$PWD/gpurun_out/session_b
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_fullsize.py tests/test_gpu_hostpath.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -15
bash scripts/gpu_variants.sh "${1:-default|}"
