#!/bin/bash
# rocprofv3 kernel trace of a short bench run: where do k_abc / k_dw_adam start and end relative to each other?
out=gpurun_out/otrace
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --steps ${STEPS:-600} --warmup ${WARM:-100} --no-cpu-baseline --no-stepwise --no-peaks > $out/bench.json 2> $out/trace.log
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
ks = [k for k in ks if k[0] in ("k_abc", "k_dw_adam")]
ks.sort(key=lambda k: k[1])
abc = [k for k in ks if k[0] == "k_abc"]; dw = [k for k in ks if k[0] == "k_dw_adam"]
print("launches", len(abc), len(dw))
# steady state: the last 300 steps
A = abc[-300:]; D = dw[-300:]
per = [b[1] - a[1] for a, b in zip(A, A[1:])]
print("k_abc period median %.2f us" % (statistics.median(per) / 1e3))
print("k_abc dur median %.2f  k_dw dur median %.2f" % (statistics.median([e - s for _, s, e in A]) / 1e3, statistics.median([e - s for _, s, e in D]) / 1e3))
# per step: abc start, abc end, dw start, dw end relative to abc start
rel = []
for a in A:
    d = [x for x in D if x[1] >= a[1]]
    if not d: continue
    d = d[0]
    rel.append(((a[2] - a[1]) / 1e3, (d[1] - a[1]) / 1e3, (d[2] - a[1]) / 1e3))
for i in (0, 1, 2, 100, 101, 102):
    if i < len(rel): print("step: abc end %.2f  dw start %.2f  dw end %.2f" % rel[i])
print("medians: abc end %.2f dw start %.2f dw end %.2f" % tuple(statistics.median(c) for c in zip(*rel)))
nxt = [(b[1] - a[2]) / 1e3 for a, b in zip(A, A[1:])]
print("next abc start - this abc end: median %.2f us" % statistics.median(nxt))
PY
python3 -c "
import json; d=json.load(open('$out/bench.json')); print(d['value'], d['ms_per_step'])"
rm -rf $out/trace
