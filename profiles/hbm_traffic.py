#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected SEPARATELY, as
MI355X_MICROARCH.md prescribes): python hbm_traffic.py <fetch counter_collection.csv> <write counter_collection.csv>
                                 python hbm_traffic.py <fetch csv> <write csv> --json "<kernel name prefix>"
                                 python hbm_traffic.py <fetch csv> <write csv> --step-json C V K E1 E2
Units: the counters are in KB; FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads).  --json prints the record
bench.py attaches to its roofline block (traffic_bytes_per_launch of one kernel); --step-json prints the training step launch by
launch (dispatches in order, a step = everything up to and including k_reduce; averaged position by position over the steps that
have the most common launch sequence) -- what bench.py attaches to roofline_step as `traffic`."""
import collections
import csv
import json
import sys



def step_records(path, name):
    """[(kernel, [values per step])] in launch order, from one pass's counter_collection.csv."""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    by_dispatch = collections.OrderedDict()   # a dispatch may appear on several rows (one per counter instance): sum them
    for r in rows:
        k = int(r["Dispatch_Id"])
        by_dispatch.setdefault(k, [r["Kernel_Name"].split("(")[0].replace("void ", ""), 0.0])[1] += float(r["Counter_Value"])
    steps, cur, live = [], [], False
    for kern, val in by_dispatch.values():   # a step: k_embed_fwd ... up to its last launch (k_reduce; k_fold_grads behind it in profiles of earlier builds)
        if live and (kern.startswith("k_embed_fwd") or not kern.startswith("k_")):
            steps.append(cur)
            live = False
        if kern.startswith("k_embed_fwd"):
            cur, live = [], True
        if live:
            cur.append((kern, val))
    if live:
        steps.append(cur)
    steps = [s for s in steps if s[-1][0].startswith(("k_fold_grads", "k_reduce"))]
    if not steps:
        return []
    common = collections.Counter(tuple(k for k, _ in s) for s in steps).most_common(1)[0][0]
    same = [s for s in steps if tuple(k for k, _ in s) == common]
    return [(kern, [s[i][1] for s in same]) for i, kern in enumerate(common)]


if len(sys.argv) > 3 and sys.argv[3] == "--step-json":
    fetch, write = step_records(sys.argv[1], "FETCH_SIZE"), step_records(sys.argv[2], "WRITE_SIZE")
    assert [k for k, _ in fetch] == [k for k, _ in write], "the two passes saw different launch sequences"
    launches = [{"kernel": k, "fetch_bytes": 2 * 1024 * sum(f) / len(f), "write_bytes": 1024 * sum(w) / len(w), "steps_averaged": len(f)}
                for (k, f), (_, w) in zip(fetch, write)]
    print(json.dumps({"dims": [int(x) for x in sys.argv[4:9]], "launches": launches,
                      "traffic_bytes_per_step": sum(r["fetch_bytes"] + r["write_bytes"] for r in launches),
                      "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                              "(gfx950 reports half of wide coalesced reads); counter units KB"}, indent=1))
    sys.exit(0)

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
