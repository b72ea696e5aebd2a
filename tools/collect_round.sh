#!/bin/bash
# Everything profiles/README.md cites for this round, in one run on the GPU box: bash tools/collect_round.sh [outdir] [round tag, e.g. r03]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r03}; mkdir -p $O
# 1. HBM traffic first: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md).  bench.py attaches `traffic` to its
#    roofline blocks from profiles/${R}_hbm_traffic.json and profiles/${R}_step_hbm_traffic.json, so these are written (on this
#    box's copy of the repo; tools/publish_round.sh copies them into the tracked tree) BEFORE the bench line is taken: line and
#    counters then come from the same build on the same box.
R=${2:-r03}
python bench.py --steps 20 --warmup 5 --min-seconds 0 --no-cpu-baseline --no-roofline > $O/dims.json 2> $O/bench.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 10 --warmup 2 --min-seconds 0 --no-cpu-baseline > /dev/null 2> $O/pmc_$c.err
done
F=$(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python3 profiles/hbm_traffic.py $F $W > $O/step_hbm_traffic.txt
python3 profiles/hbm_traffic.py $F $W --json "k_seg_sum<4, false>" > $O/hbm_traffic.json
python3 profiles/hbm_traffic.py $F $W --step-json $(python3 -c "import json; c = json.load(open('$O/dims.json'))['config']; print(c['n_cons'], c['n_vars'], c['n_cuts'], c['n_cons_edges'], c['n_cut_edges'])") > $O/step_hbm_traffic.json
cp $O/hbm_traffic.json profiles/${R}_hbm_traffic.json; cp $O/step_hbm_traffic.json profiles/${R}_step_hbm_traffic.json
echo "pmc done"
# 2. the default bench line (roofline, roofline_step, cpu_baseline)
python bench.py > $O/bench.json 2>> $O/bench.err
echo "bench done"
# 3. kernel statistics + timeline of the same workload, INCLUDING the standalone scatter-sum loop of the roofline block, so that
#    roofline.frac can be recomputed from the tracked summary (k_seg_sum row: 212,096,000 B / its average duration / 8.0e12)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 50 --warmup 10 --min-seconds 0 --no-cpu-baseline > $O/prof.json 2> $O/prof.err
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $f $O/step_kernel_stats.csv
python3 profiles/summarize.py $f 84 > $O/step_summary.txt        # 50 timed + 10 warm-up + 1 launch-count + 3 + 20 roofline_step steps
python3 profiles/timeline.py $(find $O/prof -name '*kernel_trace.csv' | head -1) > $O/step_timeline.txt
echo "stats done"
# 4. the other BASELINE shapes: kernel statistics and bench lines; the per-GPU batch sweep
bash tools/collect_cfg_profiles.sh $O "capfac 32" "indset 64" "combauc 32" > /dev/null
bash tools/batch_sweep.sh $O/batch_sweep.jsonl > /dev/null
python3 tools/sweep_table.py $O/batch_sweep.jsonl > $O/batch_sweep.txt
echo "configs done"
# 5. SQ counters, the data-parallel path rehearsed at world size 1 (weak and strong), the hipGraph replay, single-state latency,
#    epoch throughput (plain, and once under rocprofv3: the store path's kernels per batch)
bash tools/collect_sq.sh $O > $O/sq_pmc.txt 2>&1
GCNN_FORCE_DP=1 python bench.py --no-cpu-baseline --no-roofline > $O/dp_world1.json 2>> $O/bench.err
GCNN_FORCE_DP=1 python bench.py --no-cpu-baseline --no-roofline --scaling strong > $O/dp_world1_strong.json 2>> $O/bench.err
python bench.py --graph --no-cpu-baseline --no-roofline > $O/graph_replay.json 2>> $O/bench.err
python tools/latency.py > $O/single_sample_latency.txt 2>> $O/bench.err
python tools/epoch_throughput.py > $O/epoch_throughput.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_epoch -- python3 tools/epoch_throughput.py 64 40 > $O/epoch_profiled.txt 2> $O/epoch_profiled.err
python3 profiles/summarize.py $(find $O/prof_epoch -name '*kernel_stats.csv' | head -1) 1 2> /dev/null | grep -E "^kernel |k_collate|k_ranking|k_embed_fwd|k_reduce|k_wgrad" > $O/epoch_store_kernels.txt || true
echo done
