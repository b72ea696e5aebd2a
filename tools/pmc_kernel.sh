#!/bin/bash
# FETCH_SIZE of the step's kernels for one setting: tools/pmc_kernel.sh <tag> [env assignments...]
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_$tag; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -- python3 bench.py --steps 10 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-roofline $AB_ARGS > /dev/null 2> $O/err_$c.txt
done
echo "== $tag"; python3 profiles/hbm_traffic.py $(find $O/FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find $O/WRITE_SIZE -name '*counter_collection.csv' | head -1) | grep -E "k_wgrad|k_reduce|k_tail"
