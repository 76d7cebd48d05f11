#!/usr/bin/env python3
"""Combine the per-kernel PMC summaries (scripts/summarize_pmc.py output of the FETCH_SIZE, WRITE_SIZE and
MFMA passes) into profiles/<tag>_pmc_traffic.json, the file bench.py reads `roofline.traffic` from.
    python scripts/make_pmc_traffic.py fetch.json write.json mfma.json lift_b256 > profiles/r02_lift_b256_pmc_traffic.json
Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): counters are KiB; gfx950 tallies a 128-B request
at 64 B, so FETCH_SIZE is doubled for the kernels that stream 16 B per lane from contiguous segments (all
step kernels); k_gather reads random 176-B rows with 64-B requests and its RAW counter matches the known
byte count, so it is not corrected.  WRITE_SIZE is exact."""
import json
import sys

fetch, write, mfma = (json.load(open(p)) for p in sys.argv[1:4])
workload = sys.argv[4] if len(sys.argv) > 4 else "lift_b256"      # bench.py looks its workload's summary up by this tag
out = {"_note": __doc__.split("Corrections", 1)[1].strip().replace("\n", " "), "workload": workload, "kernels": {}}
for k in sorted(fetch):
    name = k.split("<")[0].split("::")[-1]
    f = fetch[k].get("FETCH_SIZE", {}).get("mean_per_launch")
    w = write.get(k, {}).get("WRITE_SIZE", {}).get("mean_per_launch")
    if f is None or w is None:
        continue
    fb = f * 1024 * (1 if name == "k_gather" else 2)
    wb = w * 1024
    e = dict(fetch_bytes=round(fb), write_bytes=round(wb), traffic_bytes_per_launch=round(fb + wb),
             launches=fetch[k]["FETCH_SIZE"]["launches"])
    for c, v in mfma.get(k, {}).items():
        e[c] = round(v["mean_per_launch"])
    out["kernels"].setdefault(name, e)
print(json.dumps(out, indent=1))
