for f in ${FILLS:-4096 2560 6144 4096}; do
  GSR_EXTRA_FLAGS="-DGSR_BWD_FILL=$f" python gaussian-pcloud-render_amd/build.py --force > /dev/null 2>&1
  python bench.py --no-cpu-baseline --no-per-view 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_frame']; print('fill $f', d['value'], 'bwd', k['render_backward'], 'roofline avg_ms', d['roofline']['avg_ms'])"
done
