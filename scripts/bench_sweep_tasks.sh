#!/bin/bash
# The eight tasks of BASELINE.json configs[4] (one per GPU there), one after the other on ONE GPU: per-task steps/s at
# batch 256.  The 8-GPU job's aggregate is 8 * K / (the slowest task's time): the last line says what that would be.
out=gpurun_out/sweep_tasks.jsonl
: > $out
for t in Lift Door Stack Wipe PickPlaceCan NutAssemblyRound TwoArmPegInHole TwoArmHandoff; do
  timeout -k 10 200 python bench.py --task $t --no-cpu-baseline --no-stepwise --no-peaks --steps 1000 --warmup 100 >> $out 2>/dev/null || exit 1
done
python3 - <<PY
import json
rows = [json.loads(l) for l in open("$out")]
slow = max(r["ms_per_step"] for r in rows)
for r in rows:
    t = r["config"]["rank_tasks"][0]
    print(f'{r["config"]["workload"].split("-")[0]:18s} obs {r["config"]["workload"].split("obs ")[1].split(",")[0]:>9s}  {r["value"]:9.1f} steps/s  {r["ms_per_step"]*1e3:6.2f} us/step  '
          + str({k: round(v["ms"] * 1e3, 2) for k, v in r["kernels"].items() if k.startswith("k_fwd") or k == "k_dw_adam"}))
print(f"8 GPUs, one task each (no data-path collective): 8 / slowest = {8 / slow * 1e3:.0f} steps/s aggregate; sum of the per-task rates = {sum(r['value'] for r in rows):.0f}")
PY
