#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 PMC counters from a counter_collection CSV -> JSON on stdout."""
import csv
import json
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0]
            c = acc[k][row["Counter_Name"]]
            c[0] += float(row["Counter_Value"])
            c[1] += 1
print(json.dumps({k: {c: {"mean_per_launch": v[0] / v[1], "launches": v[1]} for c, v in cs.items()}
                  for k, cs in acc.items()}, indent=1))
