#!/bin/bash
# three long-loop runs + one at the driver's flags, value / ms_per_step only (scratch helper)
set -e
F="--no-cpu-baseline --no-stepwise --no-peaks"
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 2000 --warmup 200 $F 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('long', d['value'], d['ms_per_step'])"; done
python3 bench.py --gpus 1 --steps 20 --warmup 5 $F 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver', d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel_us'))"
