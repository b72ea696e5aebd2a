# k_wgrad micro-benchmark, as built and with the loads / the MFMAs of its main loop removed (bash tools/micro/run_wgrad_abl.sh)
set -e
for v in "" "-DWG_ABL_NOLOAD" "-DWG_ABL_NOMFMA"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -w $v -o /tmp/bw tools/micro/bench_wgrad.hip
  echo "== variant [$v]"; /tmp/bw 1
done
