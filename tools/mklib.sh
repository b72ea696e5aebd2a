#!/bin/bash
# build a library variant for tools/ab.sh: tools/mklib.sh <name> [git-rev] [extra hipcc flags...]
# <git-rev> (optional, e.g. HEAD): build the kernel sources of that commit instead of the working tree
set -e
name=$1; shift
src=gcnn-cut-selector_amd/csrc
if [ -n "$1" ] && git rev-parse --verify -q "$1^{commit}" > /dev/null 2>&1; then
  rev=$1; shift
  tmp=$(mktemp -d); mkdir -p $tmp/gcnn-cut-selector_amd $tmp/include
  git archive $rev gcnn-cut-selector_amd/csrc include | tar -x -C $tmp
  src=$tmp/gcnn-cut-selector_amd/csrc
fi
mkdir -p tools/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -w "$@" -o tools/ab/lib_$name.so $src/gcnn_capi.hip
echo "built tools/ab/lib_$name.so from $src"
