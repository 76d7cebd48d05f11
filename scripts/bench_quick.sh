# three default-length runs (2000 steps) without the extras: value, per-kernel execution and boundary
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-stepwise 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items() if k.startswith('k_fwd') or k=='k_dw_adam'}, r['dispatch_boundary_ms'], r['frac'])"
done
