"""Do the step kernels run slower while the background index draw / gather of the next chunk is on the chip?
usage: python scratch/overlap_cost.py <kernel_trace.csv>   (rocprofv3 --kernel-trace of a long loop)"""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
bg = sorted((s, e, n) for n, s, e in ks if n in ("k_mt_randint", "k_gather"))
def overlapping(s, e):
    return [n for (bs, be, n) in bg if bs < e and s < be]
for name in ("k_abc", "k_dw_adam"):
    clean, mt, ga = [], [], []
    for n, s, e in ks:
        if n != name: continue
        o = overlapping(s, e)
        (mt if "k_mt_randint" in o else ga if "k_gather" in o else clean).append(e - s)
    for lab, v in (("alone", clean), ("with k_mt_randint", mt), ("with k_gather", ga)):
        if v: print(f"{name:10s} {lab:18s} n={len(v):5d} mean {sum(v)/len(v)/1e3:6.2f} us  median {statistics.median(v)/1e3:6.2f}")
print("background kernels:", [(n, round((e - s) / 1e3, 1)) for s, e, n in bg][-12:])
# drift along the run: mean duration and mean start-to-start period per 500 consecutive k_abc launches
abc = sorted((s, e) for n, s, e in ks if n == "k_abc")
for i in range(0, len(abc) - 500, 500):
    seg = abc[i:i + 500]
    per = [b[0] - a[0] for a, b in zip(seg, seg[1:]) if b[0] - a[0] < 100_000]
    print(f"k_abc launches {i:5d}..: duration {sum(e - s for s, e in seg)/len(seg)/1e3:6.2f} us  period {sum(per)/max(1,len(per))/1e3:6.2f} us")
