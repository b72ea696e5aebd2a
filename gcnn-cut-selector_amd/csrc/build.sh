#!/bin/sh
# Build libgcnn_hip.so for MI355X (gfx950).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
exec /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -o libgcnn_hip.so gcnn_capi.hip "$@"
