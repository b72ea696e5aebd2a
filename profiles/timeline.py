#!/usr/bin/env python3
"""Print the kernel timeline of one training step from a rocprofv3 kernel_trace.csv: python timeline.py <csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_fold_grads")]   # the last launch of a training step
if not idx:
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_reduce")]
# a step from the middle of the timed region (the run ends with the per-launch-event steps of bench.py's roofline_step, whose
# event records put gaps between the launches)
a, b = idx[min(30, len(idx) - 2)], idx[min(31, len(idx) - 1)]
prev_end = int(rows[a]["End_Timestamp"])
for r in rows[a + 1:b + 1]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][:56]:56s} dur={(en - st) / 1000:7.2f}us gap={(st - prev_end) / 1000:6.2f} grid={r['Grid_Size_X']:>7s} wg={r['Workgroup_Size_X']}")
    prev_end = en
print("step span us", (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1000)
