#!/bin/bash
# rocprofv3 passes of one round on the GPU box -> gpurun_out/prof_<tag>/ (summaries to be copied to profiles/).
#   bash scripts/profile_round.sh r02_lift_b256 [task index in parallel.SWEEP | TwoArmLift] [batch]
# Counters are collected in their own passes (never together with sys/hip/hsa traces).
set -e
tag=${1:?tag}
task=${2:-Lift}
batch=${3:-256}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
wl=$(python3 -c "print('${task}'.lower() + '_b${batch}')")
# (the fused loop only: --no-stepwise keeps the stepwise data point, whose steps interleave with per-batch gathers, out of the averages)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --task $task --batch $batch --no-cpu-baseline --no-stepwise --no-peaks > $out/bench_under_rocprof.json 2> $out/trace.log
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  set -- $pass
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 scripts/profile_loop.py --steps 512 --task $task --batch $batch > $out/$name.log 2>&1
  python3 scripts/summarize_pmc.py $(find $out/$name -name "*counter_collection.csv") > $out/pmc_$name.json
done
python3 scripts/make_pmc_traffic.py $out/pmc_fetch.json $out/pmc_write.json $out/pmc_mfma.json $wl > $out/pmc_traffic.json
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace $out/fetch $out/write $out/mfma
python3 bench.py --task $task --batch $batch > $out/bench_default.json 2> $out/bench_default.err
tail -c 400 $out/bench_default.json
