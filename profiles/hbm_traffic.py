#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected SEPARATELY, as
MI355X_MICROARCH.md prescribes): python hbm_traffic.py <fetch counter_collection.csv> <write counter_collection.csv>
Units: the counters are in KB; FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads)."""
import collections
import csv
import sys

res = collections.OrderedDict()
for path, name in ((sys.argv[1], "FETCH_SIZE"), (sys.argv[2], "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            agg[r["Kernel_Name"].split("(")[0][:34] + " grid=" + r["Grid_Size"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res.setdefault(k, {})[name] = sum(v) / len(v)
print(f"{'kernel (grid; launches of equal grid are averaged)':56s} {'fetch MB':>10s} {'write MB':>10s}")
for k, v in res.items():
    if k.startswith("k_") or k.startswith("void k_"):
        print(f"{k:56s} {v.get('FETCH_SIZE', 0) * 2 * 1024 / 1e6:10.1f} {v.get('WRITE_SIZE', 0) * 1024 / 1e6:10.1f}")
