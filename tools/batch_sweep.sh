#!/bin/bash
# Per-GPU batch sizes the 8-GPU BASELINE configs imply (configs[3]/[4]: 4 / 8 samples per GPU) and the setcov sweep around
# the headline batch: bash tools/batch_sweep.sh [out.jsonl]   (one bench line per point: ms/step, edges/s, launches)
set -e
cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/batch_sweep.jsonl}; mkdir -p $(dirname $O); : > $O
for cfg in "setcov 4" "setcov 8" "setcov 16" "setcov 32" "setcov 64" "setcov 128" "capfac 4" "indset 8" "combauc 32" "capfac 32" "indset 64"; do
  set -- $cfg
  python bench.py --problem $1 --batch $2 --no-cpu-baseline --no-roofline >> $O 2>> ${O%.jsonl}.err
  echo "$cfg done"
done
