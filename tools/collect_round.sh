#!/bin/bash
# Everything profiles/README.md cites for this round, in one run on the GPU box: bash tools/collect_round.sh [outdir]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r02}; mkdir -p $O
steps() { python3 - "$1" <<'PY'
import json, sys
print(json.load(open(sys.argv[1]))["timing"]["steps_timed"] + json.load(open(sys.argv[1]))["warmup"])
PY
}
# 1. the default bench line (roofline, roofline_step, cpu_baseline)
python bench.py > $O/bench.json 2> $O/bench.err
# 2. kernel statistics + timeline of the same workload
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 50 --warmup 10 --min-seconds 0 --no-cpu-baseline --no-roofline > $O/prof.json 2> $O/prof.err
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $f $O/step_kernel_stats.csv
python3 profiles/summarize.py $f 60 > $O/step_summary.txt        # 50 timed + 10 warm-up steps
python3 profiles/timeline.py $(find $O/prof -name '*kernel_trace.csv' | head -1) > $O/step_timeline.txt
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 10 --warmup 2 --min-seconds 0 --no-cpu-baseline > /dev/null 2> $O/pmc_$c.err
done
python3 profiles/hbm_traffic.py $(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1) > $O/step_hbm_traffic.txt
python3 profiles/hbm_traffic.py $(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1) --json "k_seg_sum<4, false>" > $O/hbm_traffic.json
# 4. the other BASELINE shapes: kernel statistics
bash tools/collect_cfg_profiles.sh $O "capfac 32" "indset 64" "combauc 32" > /dev/null
for cfg in "combauc 32" "capfac 32" "indset 64"; do set -- $cfg; python bench.py --problem $1 --batch $2 --no-cpu-baseline --no-roofline >> $O/other_configs.jsonl 2>> $O/bench.err; done
# 5. SQ counters, the data-parallel path rehearsed at world size 1 (weak and strong), the hipGraph replay, single-state latency, epoch throughput
bash tools/collect_sq.sh $O > $O/sq_pmc.txt 2>&1
GCNN_FORCE_DP=1 python bench.py --no-cpu-baseline --no-roofline > $O/dp_world1.json 2>> $O/bench.err
GCNN_FORCE_DP=1 python bench.py --no-cpu-baseline --no-roofline --scaling strong > $O/dp_world1_strong.json 2>> $O/bench.err
python bench.py --graph --no-cpu-baseline --no-roofline > $O/graph_replay.json 2>> $O/bench.err
python tools/latency.py > $O/single_sample_latency.txt 2>> $O/bench.err
python tools/epoch_throughput.py > $O/epoch_throughput.txt 2>&1
echo done
