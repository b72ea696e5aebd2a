#!/bin/bash
# interleaved A/B of library builds on one box
for i in 1 2 3; do
  for v in "$@"; do
    r=$(GCNN_LIB=$PWD/tools/ab/lib_$v.so python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline $AB_ARGS 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$v $r"
  done
done
