for i in 1 2; do
for e in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_ENABLE_INTERRUPT=0" "AMD_DIRECT_DISPATCH=1"; do
  for b in 32 4; do
    r=$(env $e python bench.py --batch $b --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "[$e] x$b $r"
  done
done
done
