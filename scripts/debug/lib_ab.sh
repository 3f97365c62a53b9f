# A/B of two builds of the library in one lease: GSR_LIB=<other .so> against the in-tree one, alternating
OTHER=${1:?path of the other libgsr_hip.so}
for rep in 1 2; do
for lib in "$OTHER" ""; do
  for vpc in 12 1; do
  GSR_LIB=$lib python bench.py --no-cpu-baseline --no-per-view --views-per-call $vpc 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_frame']; print('lib [%s] vpc $vpc' % ('$lib' or 'in-tree'), d['value'], round(sum(k.values()),4), 'pre', k['preprocess'], 'fwd', k['render_forward'], 'bwd', k.get('render_backward'), 'pbwd', k.get('preprocess_backward'))"
  done
done
done
