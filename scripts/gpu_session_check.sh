set -o pipefail
OUT=gpurun_out/${1:-r2s}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; rc=$?; echo "gpu tests rc=$rc" | tee $OUT/rc.txt
tail -5 $OUT/gputests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_20.json 2> $OUT/bench_20.err; echo "bench20 rc=$?"
timeout -k 10 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
cat $OUT/bench_20.json $OUT/bench_default.json
