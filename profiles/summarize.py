#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv per training step: python summarize.py <csv> <steps>
`steps` = number of training steps the profiled run executed (warm-up included).  Only kernels of the training step are
summed: plan kernels, the standalone scatter-sum pass (K9) and the inference-only edge variant of the roofline_fused bench are
listed but not counted."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
NOT_STEP = ("seg_sum", "seg_bcast", "iota", "gather_edges", "seg_offsets", "check_edges", "k_edge_fwd<4, false>",
            "k_edge_fwd<2, false>", "k_edge_fwd<1, false>", "k_edge_fwd_block<false>", "k_edge_fwd_long<false>", "k_iplan", "k_infer",
            "k_rank_scores", "k_collate", "k_stats", "k_expand")
tot = 0.0
print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'us/step':>9s}")
for r in rows:
    t = int(r["TotalDurationNs"])
    name = r["Name"]
    step = name.startswith(("k_", "void k_")) and not any(x in name for x in NOT_STEP)
    print(f"{name[:72]:72s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1000:9.2f} {t / steps / 1000:9.2f}{'' if step else '   (not a step kernel)'}")
    if step:
        tot += t / steps / 1000
print(f"training-step kernels: {tot:.1f} us/step over {steps} steps")
