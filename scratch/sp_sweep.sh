#!/bin/bash
for cfg in "256 4" "256 2" "512 4" "512 2" "512 1" "1024 2" "1024 1"; do
  set -- $cfg
  SAC_FORCE_SP=$2 python bench.py --batch $1 --no-cpu-baseline --no-stepwise --steps 800 --warmup 100 > /tmp/sp.json || exit 1
  python - <<PY
import json
d = json.load(open("/tmp/sp.json"))
print("B=$1 SP=$2", d["value"], {k: v["ms"] for k, v in d["kernels"].items() if k[2] in "fbd"})
PY
done
