#!/bin/bash
# kernel-trace averages of the default bench (scratch helper): prints kernel_stats.csv head
set -e
out=gpurun_out/qtrace
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py "$@" --no-cpu-baseline --no-stepwise --no-peaks > $out/bench.json 2> $out/trace.log
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace
head -6 $out/kernel_stats.csv | cut -d, -f1-8
