#!/bin/bash
# A/B of library builds on ONE box, alternating: scratch/ab/<name>.so copied over the package's library in turn, REPS rounds of the
# long bench each (steps/s), then one rocprofv3 kernel-trace per build (average launch durations of the step kernels).
REPS=${REPS:-3}
cp robosuite_benchmark_amd/libsac_hip.so /tmp/keep.so
for r in $(seq $REPS); do
  for so in scratch/ab/*.so; do
    cp $so robosuite_benchmark_amd/libsac_hip.so
    python3 bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline --no-stepwise --no-peaks "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so', d['value'], d['ms_per_step'])"
  done
done
for so in scratch/ab/*.so; do
  cp $so robosuite_benchmark_amd/libsac_hip.so
  bash scratch/quick_trace.sh "$@" > /dev/null 2>&1
  python3 - "$so" <<'P'
import csv, sys
rows = list(csv.DictReader(open("gpurun_out/qtrace/kernel_stats.csv")))
print(sys.argv[1], " ".join(f"{r['Name'].split('(')[0].split('::')[-1][:12]}={float(r['AverageNs'])/1000:.2f}us" for r in rows[:2]))
P
done
cp /tmp/keep.so robosuite_benchmark_amd/libsac_hip.so
