#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv per training step: python summarize.py <csv> <steps>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tot = 0.0
print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'us/step':>9s}")
for r in rows:
    t = int(r["TotalDurationNs"])
    name = r["Name"]
    print(f"{name[:72]:72s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1000:9.2f} {t / steps / 1000:9.2f}")
    if name.startswith(("k_", "void k_")) and "seg_sum" not in name and "iota" not in name and "gather_edges" not in name \
            and "seg_offsets" not in name:
        tot += t / steps / 1000
print(f"step kernels (k_* except plan/K9 benches): {tot:.1f} us/step")
