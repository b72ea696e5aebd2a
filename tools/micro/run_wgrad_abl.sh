set -e
for v in "" "-DWG_ABL_NOLOAD" "-DWG_ABL_NOMFMA"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -w $v -o /tmp/bw tools/micro/bench_wgrad.hip
  echo "== variant [$v] zeros"; /tmp/bw 0
  echo "== variant [$v] random"; /tmp/bw 1
done
