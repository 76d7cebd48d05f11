#!/bin/bash
set -e
out=gpurun_out/calltrace
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 scratch/short_loop.py 20 > $out/short.log 2> $out/trace.log
cp $(find $out/trace -name "*kernel_trace.csv" | head -1) $out/kernel_trace.csv
rm -rf $out/trace
cat $out/short.log
python3 scratch/call_timeline.py $out/kernel_trace.csv > $out/timeline.txt
head -80 $out/timeline.txt
