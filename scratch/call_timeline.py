"""Timeline of ONE short sac_train_loop call from a rocprofv3 --kernel-trace CSV of scratch/short_loop.py: every kernel
of the call with its start / end relative to the call's first kernel.  usage: python scratch/call_timeline.py <kernel_trace.csv> [group]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:28], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows)
groups, cur = [], [ks[0]]
for k in ks[1:]:
    if k[0] - cur[-1][1] > 1_000_000: groups.append(cur); cur = [k]
    else: cur.append(k)
groups.append(cur)
print("groups:", [len(g) for g in groups])
gi = int(sys.argv[2]) if len(sys.argv) > 2 else -4
g = groups[gi]
t0 = g[0][0]
prev_end = t0
for s, e, n, q in g:
    print(f"{(s - t0)/1e3:8.2f} -> {(e - t0)/1e3:8.2f}  ({(e - s)/1e3:6.2f} us, idle before {(s - prev_end)/1e3:6.2f})  q{q}  {n}")
    prev_end = max(prev_end, e)
