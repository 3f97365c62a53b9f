for sh in ${SHIFTS:-9 10 11}; do
  GSR_EXTRA_FLAGS="-DGSR_BWD_CHUNK_SHIFT=$sh" python gaussian-pcloud-render_amd/build.py --force > /dev/null 2>&1
  for vpc in ${VPCS:-12 1}; do
  python bench.py --no-cpu-baseline --no-per-view --views-per-call $vpc 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shift $sh vpc $vpc', d['value'], d['kernels_ms_per_frame']['render_backward'], d['kernels_ms_per_frame']['render_forward'], d['kernels_ms_per_frame']['bwd_items'])"
  done
done
