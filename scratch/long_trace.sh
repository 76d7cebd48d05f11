#!/bin/bash
set -e
out=gpurun_out/longtrace
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --steps 4000 --warmup 200 --no-cpu-baseline --no-stepwise --no-peaks > $out/bench.json 2> $out/trace.log
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 scratch/overlap_cost.py $f
rm -rf $out/trace
