#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected SEPARATELY, as
MI355X_MICROARCH.md prescribes): python hbm_traffic.py <fetch counter_collection.csv> <write counter_collection.csv>
                                 python hbm_traffic.py <fetch csv> <write csv> --json "<kernel name prefix>"
Units: the counters are in KB; FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads).  --json prints the record
bench.py attaches to its roofline block (traffic_bytes_per_launch of one kernel)."""
import collections
import csv
import json
import sys

res = collections.OrderedDict()
for path, name in ((sys.argv[1], "FETCH_SIZE"), (sys.argv[2], "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            agg[r["Kernel_Name"].split("(")[0][:34] + " grid=" + r["Grid_Size"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res.setdefault(k, {})[name] = sum(v) / len(v)
        res[k]["launches_" + name] = len(v)
if len(sys.argv) > 4 and sys.argv[3] == "--json":
    for k, v in res.items():
        if k.startswith("void " + sys.argv[4]) or k.startswith(sys.argv[4]):
            print(json.dumps({"kernel": k, "fetch_size_kb": v.get("FETCH_SIZE", 0.0), "write_size_kb": v.get("WRITE_SIZE", 0.0),
                              "launches": v.get("launches_FETCH_SIZE", 0),
                              "traffic_bytes_per_launch": (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024,
                              "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (bench.py --steps 10); FETCH_SIZE doubled "
                                      "per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); units KB"}, indent=1))
            break
    sys.exit(0)
print(f"{'kernel (grid; launches of equal grid are averaged)':56s} {'fetch MB':>10s} {'write MB':>10s}")
for k, v in res.items():
    if k.startswith("k_") or k.startswith("void k_"):
        print(f"{k:56s} {v.get('FETCH_SIZE', 0) * 2 * 1024 / 1e6:10.1f} {v.get('WRITE_SIZE', 0) * 1024 / 1e6:10.1f}")
