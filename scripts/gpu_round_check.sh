set -o pipefail
mkdir -p gpurun_out/r2a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/gputests.log 2>&1; echo "gpu tests rc=$?" | tee gpurun_out/r2a/rc.txt
tail -5 gpurun_out/r2a/gputests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2a/bench_20.json 2> gpurun_out/r2a/bench_20.err; echo "bench20 rc=$?"
timeout -k 10 400 python bench.py > gpurun_out/r2a/bench_default.json 2> gpurun_out/r2a/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ("bench_20","bench_default"):
    try:
        d=json.load(open(f"gpurun_out/r2a/{f}.json"))
        print(f, d["value"], d["ms_per_step"], d.get("fixed_call_us"), d.get("short_loop"), d["roofline"]["frac"], d.get("peaks_measured"), d["roofline_gather"]["K1"], d["roofline_gather"]["K1000"], d.get("ingest"), d.get("cpu_baseline",{}) and d["cpu_baseline"].get("value"))
        print({k:v["ms"] for k,v in d["kernels"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
