for plan in "4,12,48,192" "2,14,48,192" "1,3,12,48,192" "8,8,48,192"; do
  echo "plan $plan"
  for i in 1 2 3; do
    SAC_CHUNK_PLAN=$plan timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-stepwise --no-peaks 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  ', d['value'], d['timed_region_us'], d['short_loop']['value'])"
  done
done
