#!/bin/bash
# copy the files of a collect_round.sh run that profiles/README.md cites into profiles/ (tracked): tools/publish_round.sh gpurun_out/r02 r02
set -e
S=$1; R=$2
cp $S/bench.json profiles/${R}_bench.json
cp $S/step_kernel_stats.csv profiles/${R}_step_kernel_stats.csv
cp $S/step_summary.txt profiles/${R}_step_summary.txt
cp $S/step_timeline.txt profiles/${R}_step_timeline.txt
cp $S/hbm_traffic.json profiles/${R}_hbm_traffic.json
cp $S/step_hbm_traffic.txt profiles/${R}_step_hbm_traffic.txt
[ -f $S/step_hbm_traffic.json ] && cp $S/step_hbm_traffic.json profiles/${R}_step_hbm_traffic.json
[ -f $S/batch_sweep.jsonl ] && cp $S/batch_sweep.jsonl profiles/${R}_batch_sweep.jsonl && cp $S/batch_sweep.txt profiles/${R}_batch_sweep.txt
[ -f $S/epoch_store_kernels.txt ] && cp $S/epoch_store_kernels.txt profiles/${R}_epoch_store_kernels.txt
cp $S/sq_pmc.txt profiles/${R}_sq_pmc.txt
[ -f $S/sq_pmc_capfac32.txt ] && cp $S/sq_pmc_capfac32.txt profiles/${R}_sq_pmc_capfac32.txt
for c in capfac32 indset64 combauc32; do cp $S/${c}_summary.txt profiles/${R}_${c}_summary.txt; done
[ -f $S/other_configs.jsonl ] && cp $S/other_configs.jsonl profiles/${R}_other_configs.jsonl
cp $S/dp_world1.json profiles/${R}_dp_world1.json
cp $S/dp_world1_strong.json profiles/${R}_dp_world1_strong.json
cp $S/graph_replay.json profiles/${R}_graph_replay.json
cp $S/single_sample_latency.txt profiles/${R}_single_sample_latency.txt
cp $S/epoch_throughput.txt profiles/${R}_epoch_throughput.txt

echo published
