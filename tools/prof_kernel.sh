#!/bin/bash
# average duration of the step's kernels under rocprofv3 for one setting: tools/prof_kernel.sh <tag> [env assignments...]
# (the env assignments are exported in this shell: rocprofv3 must start python directly)
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pk_$tag; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 50 --warmup 10 --min-seconds 0 --no-cpu-baseline --no-roofline $AB_ARGS > $O/bench.json 2> $O/err.txt
python3 profiles/summarize.py $(find $O -name '*kernel_stats.csv' | head -1) 60 | grep -v "not a step kernel" > $O/summary.txt
echo "== $tag"; grep -E "k_wgrad|k_reduce|training-step" $O/summary.txt
