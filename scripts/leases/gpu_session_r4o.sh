#!/usr/bin/env bash
python gaussian-pcloud-render_amd/build.py > /dev/null 2>&1
timeout 600 python scripts/debug/overlap_warm.py 2>&1 | grep -v amdgpu.ids
