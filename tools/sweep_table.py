"""bench lines (jsonl) -> one row per point: python tools/sweep_table.py <file.jsonl>"""
import json, sys
print(f"{'problem':8s} {'batch':>5s} {'edges/step':>11s} {'ms/step':>8s} {'G edges/s':>10s} {'launches':>8s}")
for ln in open(sys.argv[1]):
    if not ln.strip().startswith("{"): continue
    d = json.loads(ln); c = d["config"]
    print(f"{c.get('problem', '?'):8s} {c.get('batch_per_gpu', c.get('global_batch_samples', 0)):5d} {int(c['edges_per_step']):11d} "
          f"{d['ms_per_step']:8.4f} {d['value'] / 1e9:10.3f} {c.get('library_launches_per_step', 0):8d}")
