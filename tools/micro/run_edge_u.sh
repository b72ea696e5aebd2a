set -e
for u in "4 4" "8 4" "16 8" "8 8" "2 2"; do set -- $u
  hipcc -O3 --offload-arch=gfx950 -w -DEDGE_U=$1 -DEDGE_UB=$2 -o /tmp/be tools/micro/bench_edge.hip
  echo "== EDGE_U=$1 EDGE_UB=$2"; /tmp/be | grep -E "slots=4|dir=1 slots=2" | grep -v "check\|r01"
done
