set -e
hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -o /tmp/bw tools/micro/bench_wgrad.hip
/tmp/bw 1
