#!/usr/bin/env bash
# round 5, lease I: depth hints -- the new tests, then the suite's core (the sort kernels were touched)
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hints.py -x -q -m gpu -s 2>&1 | grep -v amdgpu.ids | tail -30

