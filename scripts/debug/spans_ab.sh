# A/B in one lease: bounding-box clipping only (-DGSR_NO_SPANS) against row spans, 12 views per call and one view per call
for flag in "-DGSR_NO_SPANS" "" "-DGSR_NO_SPANS" ""; do
  GSR_EXTRA_FLAGS="$flag" python gaussian-pcloud-render_amd/build.py --force > /dev/null 2>&1
  for vpc in 12 1; do
  python bench.py --no-cpu-baseline --no-per-view --views-per-call $vpc 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_frame']; print('flags [$flag] vpc $vpc', d['value'], d['config']['list_pairs_avg'], round(sum(k.values()),4), k)"
  done
done
