set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01b; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/prof.log 2>&1
for cfg in "combauc 32" "capfac 4" "capfac 32" "indset 64"; do set -- $cfg; python bench.py --problem $1 --batch $2 --no-cpu-baseline --no-roofline >> $O/other.jsonl 2>> $O/bench.err; done
GCNN_FORCE_DP=1 python bench.py --no-cpu-baseline --no-roofline > $O/dp1.json 2>> $O/bench.err
python bench.py --graph --no-cpu-baseline --no-roofline > $O/graph.json 2>> $O/bench.err
python tools/epoch_throughput.py > $O/epoch.log 2>&1
python tools/latency.py > $O/latency.log 2>&1 || true
echo done
