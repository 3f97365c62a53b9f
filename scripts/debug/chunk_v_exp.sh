for th in ${THS:-4 64}; do
  GSR_EXTRA_FLAGS="-DGSR_CHUNK_V=$th" python gaussian-pcloud-render_amd/build.py --force > /dev/null 2>&1
  for vpc in ${VPCS:-4 6 8}; do
  python bench.py --no-cpu-baseline --no-per-view --views-per-call $vpc 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_frame']; print('1024 from V >= $th; vpc $vpc', d['value'], 'bwd', k['render_backward'], 'items', k['bwd_items'], 'fwd', k['render_forward'])"
  done
done
