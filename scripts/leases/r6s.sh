#!/usr/bin/env bash
# round 6, lease S: test of the short host path (equality with the general path, retry inside it)
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_hostpath.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -8
