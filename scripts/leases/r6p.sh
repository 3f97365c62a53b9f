#!/usr/bin/env bash
# round 6, lease P: split extra-channel layout (xyz + hit shared, normals per view): tests of the channel paths, the call-trace replay, timing of render_passes
set -u
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x -k "batch or passes or api or integration or cabi" 2>&1 | grep -v amdgpu.ids | tail -2
python scripts/bench_passes.py 2>&1 | grep -v amdgpu.ids | tail -6
