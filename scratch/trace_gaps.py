"""Gaps between consecutive step kernels from a rocprofv3 --kernel-trace CSV (dispatch start/end timestamps).
usage: python scratch/trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
ks = [k for k in ks if k[0] in ("k_abc", "k_dw_adam", "k_fwd_a", "k_fwd_b", "k_bwd")]
ks.sort(key=lambda k: k[1])
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for i, (n, s, e) in enumerate(ks):
    dur[n].append(e - s)
    if i + 1 < len(ks):
        gap[n + " -> " + ks[i + 1][0]].append(ks[i + 1][1] - e)
for n, v in dur.items():
    print(f"{n:12s} n={len(v):5d} dur median {statistics.median(v)/1e3:7.2f} us  mean {sum(v)/len(v)/1e3:7.2f}")
for n, v in gap.items():
    v = sorted(v)
    print(f"{n:24s} n={len(v):5d} gap median {statistics.median(v)/1e3:7.2f} us  p10 {v[len(v)//10]/1e3:6.2f} p90 {v[9*len(v)//10]/1e3:6.2f}")
starts = [s for n, s, e in ks if n in ("k_abc", "k_fwd_a")]
d = [b - a for a, b in zip(starts, starts[1:])]
print("step period median %.2f us" % (statistics.median(d) / 1e3))
