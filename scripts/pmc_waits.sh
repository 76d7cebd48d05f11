#!/bin/bash
# One counter pass: where do the waves of the step kernels spend their cycles?  (SQ block: WAVE_CYCLES = WAIT_ANY +
# WAIT_INST_ANY + ACTIVE_INST_ANY, quad-cycles; MFMA busy in cycles.)   bash scripts/pmc_waits.sh <task> <batch> <out dir>
set -e
task=${1:-Door}; batch=${2:-1024}; out=${3:-gpurun_out/pmc_waits}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD \
  --kernel-trace --output-format csv -d $out/p1 -- python3 scripts/profile_loop.py --steps 256 --task $task --batch $batch > $out/p1.log 2>&1
python3 scripts/summarize_pmc.py $(find $out/p1 -name "*counter_collection.csv") > $out/pmc_waits.json
rm -rf $out/p1
python3 - $out/pmc_waits.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "k_" in k and "gather" not in k and "randint" not in k:
        print(k, {c: round(x["mean_per_launch"]) for c, x in v.items()})
PY
