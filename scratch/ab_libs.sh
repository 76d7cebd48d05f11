#!/bin/bash
# A/B of library builds on ONE box: scratch/ab/<name>.so are copied over the package's library in turn, each traced
# with scratch/quick_trace.sh (kernel averages) -- box-to-box differences are larger than most kernel changes.
set -e
cp robosuite_benchmark_amd/libsac_hip.so /tmp/keep.so
for so in scratch/ab/*.so; do
  cp $so robosuite_benchmark_amd/libsac_hip.so
  echo "== $so"
  bash scratch/quick_trace.sh "$@" | sed -n 2,3p | grep -o "k_[a-z_]*\|)\",[0-9]*,[0-9]*,[0-9.]*" | paste - - 
  python3 bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline --no-stepwise --no-peaks "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('long', d['value'], d['ms_per_step'])"
done
cp /tmp/keep.so robosuite_benchmark_amd/libsac_hip.so
