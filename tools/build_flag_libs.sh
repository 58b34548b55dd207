#!/bin/bash
# Scratch builds of the whole library under one extra compiler flag set each (A/B of LLVM scheduling options):
#   tools/build_flag_libs.sh <name> <flags...>   ->  tools/scratch/lib_<name>.so
# Same per-file flags as samble_amd/csrc/Makefile.  Never shipped (tools/scratch/ is git-ignored).
set -e
name=$1; shift
cd "$(dirname "$0")/../samble_amd/csrc"
out=/tmp/flaglib_$name
mkdir -p $out ../../tools/scratch
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function"
build_one() {
  f=$1; shift
  extra=""
  case $f in select|score|chain) extra="-ffp-contract=off";; knn|knn_stream|knn_duo) extra="-fno-honor-nans";; esac
  hipcc $base $extra "$@" -c $f.hip -o $out/$f.o
}
export -f build_one; export base out
ls *.hip | sed 's/\.hip$//' | xargs -P 8 -I{} bash -c 'build_one {} '"$*"
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/lib_$name.so $out/*.o
ls -la ../../tools/scratch/lib_$name.so
