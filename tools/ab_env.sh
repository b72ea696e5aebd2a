#!/bin/bash
# interleaved A/B of environment settings on one box: tools/ab_env.sh "A=1" "A=0 B=2" ...   (each argument: env assignments)
for i in 1 2 3; do
  for v in "$@"; do
    r=$(env $v python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline $AB_ARGS 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "[$v] $r"
  done
done
