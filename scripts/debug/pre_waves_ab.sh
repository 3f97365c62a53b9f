for flag in "" "-DGSR_PRE_WAVES=4" "" "-DGSR_PRE_WAVES=4"; do
  GSR_EXTRA_FLAGS="$flag" python gaussian-pcloud-render_amd/build.py --force > /dev/null 2>&1
  for vpc in 12 1; do
  python bench.py --no-cpu-baseline --no-per-view --views-per-call $vpc 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_frame']; print('flags [$flag] vpc $vpc', d['value'], 'pre', k['preprocess'])"
  done
done
