#!/bin/bash
# interleaved A/B of tuning knobs (a -DGCNN_TUNING build) on one box: tools/ab_env.sh <lib> "<problem> <batch>" "ENV1=.. ENV2=.." "ENV.." ...
# ("-" = no knob set)
lib=$1; cfg=$2; shift 2
p=${cfg% *}; b=${cfg#* }
for i in 1 2; do
  for e in "$@"; do
    [ "$e" = "-" ] && envs="" || envs="$e"
    r=$(env $envs GCNN_LIB=$PWD/tools/ab/lib_$lib.so python bench.py --problem $p --batch $b --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$p x$b [$e] $r"
  done
done
