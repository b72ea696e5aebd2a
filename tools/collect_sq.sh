#!/bin/bash
# SQ wave-cycle breakdown per kernel of the training step (own rocprofv3 pass, counters only):
#   bash tools/collect_sq.sh <outdir> [bench.py args]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; shift; mkdir -p $O; rm -rf $O/sq   # one run per directory: the report below reads the first counter file it finds
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq -- python3 bench.py --steps 6 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-roofline "$@" > $O/sq.log 2>&1
python3 - $O <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/sq/**/*counter_collection.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:30] + " g" + r["Grid_Size"]
    agg.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':44s} {'waves':>7s} {'cyc/wave':>9s} {'wait%':>6s} {'winst%':>7s} {'active%':>8s} {'valu/wave':>10s} {'mfma_busy/wave':>15s}")
for k, c in agg.items():
    if not (k.startswith("k_") or k.startswith("void k_")): continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    w = max(m.get("SQ_WAVES", 1), 1); cyc = m.get("SQ_WAVE_CYCLES", 0) * 4
    pc = lambda n: 100 * m.get(n, 0) * 4 / max(cyc, 1)
    print(f"{k:44s} {w:7.0f} {cyc / w:9.0f} {pc('SQ_WAIT_ANY'):6.1f} {pc('SQ_WAIT_INST_ANY'):7.1f} {pc('SQ_ACTIVE_INST_ANY'):8.1f} {m.get('SQ_INSTS_VALU', 0) / w:10.0f} {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / w:15.0f}")
PY
