#!/bin/bash
# interleaved A/B of library builds on the other BASELINE shapes: ./tools/ab_cfg.sh prev new ["capfac 32" ...]
libs="$1 $2"; shift 2
[ $# -eq 0 ] && set -- "combauc 32" "capfac 4" "indset 64"
for cfg in "$@"; do
  p=${cfg% *}; b=${cfg#* }
  for i in 1 2; do
    for v in $libs; do
      r=$(GCNN_LIB=$PWD/tools/ab/lib_$v.so python bench.py --problem $p --batch $b --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
      echo "$p x$b $v $r"
    done
  done
done
