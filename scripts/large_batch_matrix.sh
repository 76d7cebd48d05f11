#!/bin/bash
# Door 46/7 at batch 1024 (BASELINE config 3) and 2048: the chained forward launch (csrc/sac_chain.h) against the
# four-launch step.  Prints value + per-kernel ms of each combination.
# usage (GPU box): bash scripts/large_batch_matrix.sh [out.txt]
out=${1:-gpurun_out/large_batch_matrix.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
run() {   # label, task, batch, env...
  label=$1; task=$2; batch=$3; shift 3
  line=$(env "$@" python bench.py --task "$task" --batch "$batch" --steps 600 --warmup 60 --buffer 200000 --no-cpu-baseline \
         --no-stepwise --no-peaks --profile-steps 300 2>/dev/null | tail -1)
  python - "$label" "$task" "$batch" "$line" >> "$out" <<'PY'
import json, sys
label, task, batch, line = sys.argv[1:5]
try:
    d = json.loads(line)
    k = d["kernels"]
    ks = "  ".join(f"{n} {k[n]['ms']*1e3:.2f}" for n in k if n.startswith("k_") and n not in ("k_gather", "k_mt_randint"))
    print(f"{task} B={batch} {label:22s} {d['value']:9.1f} steps/s  {d['ms_per_step']*1e3:6.2f} us/step   {ks}   whole-step frac {d['roofline']['whole_step']['frac']:.3f}")
except Exception as e:
    print(f"{task} B={batch} {label}: FAILED {e}: {line[:200]}")
PY
}
for cfg in "Door 1024" "Door 2048" "Wipe 1024" "TwoArmHandoff 1024"; do
  set -- $cfg
  run "four launches" $1 $2 SAC_CHAIN=0
  run "chain"         $1 $2 SAC_CHAIN=1
done
cat "$out"
