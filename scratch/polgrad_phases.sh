#!/bin/bash
# which phase of k_g_polgrad takes the time: builds with a phase compiled out (scratch/ab/skip{0,1,2}.so), rocprofv3 averages
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in 0 1 2; do
  rm -rf gpurun_out/gp/t; mkdir -p gpurun_out/gp
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gp/t -- python3 scratch/run_with.py scratch/ab/skip$v.so scripts/general_one_shape.py 512,512 512,512 256 400 > gpurun_out/gp/one.log 2>&1
  f=$(find gpurun_out/gp/t -name "*kernel_stats.csv" | head -1)
  echo "skip$v: $(grep polgrad $f | cut -d, -f1-4 | cut -c1-120)"
done
rm -rf gpurun_out/gp/t
