#!/bin/bash
# rocprofv3 kernel statistics of one training step on the BASELINE shapes other than the headline one.
# usage (GPU box): bash tools/collect_cfg_profiles.sh <outdir> ["problem batch" ...]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; shift; mkdir -p $O
[ $# -eq 0 ] && set -- "capfac 32" "indset 64" "combauc 32"
for cfg in "$@"; do
  p=${cfg% *}; b=${cfg#* }
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${p}${b} -- python3 bench.py --problem $p --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $O/${p}${b}.log 2>&1
  f=$(find $O/prof_${p}${b} -name '*kernel_stats.csv' | head -1)
  python3 profiles/summarize.py $f 35 > $O/${p}${b}_summary.txt
  echo "$cfg done"
done
