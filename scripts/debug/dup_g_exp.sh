for g in ${GS:-1 2 4}; do
  GSR_EXTRA_FLAGS="-DGSR_DUP_G=$g" python gaussian-pcloud-render_amd/build.py --force > /dev/null 2>&1
  for vpc in ${VPCS:-12 1}; do
  python bench.py --no-cpu-baseline --no-per-view --views-per-call $vpc 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_frame']; print('dup_g $g vpc $vpc', d['value'], 'dup', k['duplicate'])"
  done
done
