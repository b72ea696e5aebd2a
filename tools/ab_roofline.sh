#!/bin/bash
# interleaved A/B of library builds on one box, standalone scatter-sum pass (bench.py's roofline block): tools/ab_roofline.sh <name> ...
for i in 1 2 3; do
  for v in "$@"; do
    r=$(GCNN_LIB=$PWD/tools/ab/lib_$v.so python bench.py --steps 20 --warmup 5 --min-seconds 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read())['roofline']; print(d['us_per_launch'], d['achieved'], d['frac'])")
    echo "$v $r"
  done
done
