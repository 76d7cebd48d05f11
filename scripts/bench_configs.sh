#!/bin/bash
# Other BASELINE.json configurations through bench.py (one JSON line each) -> gpurun_out/configs_<tag>.jsonl
tag=${1:-x}
out=gpurun_out/configs_$tag.jsonl
: > $out
for cfg in "Door 1024" "Lift 512" "TwoArmLift 256" "Wipe 256" "Lift 128" "Lift 2048" "Lift 4096"; do
  set -- $cfg
  python bench.py --task $1 --batch $2 --no-cpu-baseline --steps 1000 --warmup 100 >> $out 2>> gpurun_out/configs_$tag.err || exit 1
done
python - <<PY
import json
for l in open("$out"):
    d = json.loads(l)
    print(d["config"]["workload"][:40], "|", d["value"], "steps/s", d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items() if k.startswith("k_") and k not in ("k_gather", "k_mt_randint")}, "step TFLOP/s", d["roofline"]["whole_step"]["tflops"])
PY
