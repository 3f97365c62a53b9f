#!/usr/bin/env bash
# round 5, lease B: the tests touched by the call-trace fixture, the per-view extra channels, the opt-in overlap and markVisible
set -u
OUT=$PWD/gpurun_out/r5b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_passes.py tests/test_gpu_hostpath.py tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_configs.py -x -q -m gpu -k "passes or hostpath or mark_visible or extra_channels or config0 or trace or overlap" 2>&1 | grep -v amdgpu.ids | tail -25
