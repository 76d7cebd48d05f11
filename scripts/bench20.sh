# the driver's command line, five times in fresh processes (plus the warm 20-step rate each run reports)
for i in 1 2 3 4 5; do
  timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['timed_region_us'], d['short_loop']['value'], d['fixed_call_us'])"
done
