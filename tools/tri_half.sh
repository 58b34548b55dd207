#!/bin/bash
# Timing-only: the whole library with THREE products per k-step instead of six (-DSAMBLE_TRI_HALF; results are
# bf16x3-accurate, i.e. wrong) -- what a two-plane scheme would buy the step.  Builds tools/scratch/lib_tri_half.so.
set -e
cd "$(dirname "$0")/../samble_amd/csrc"
mkdir -p ../../tools/scratch /tmp/trihalf
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -w -DSAMBLE_TRI_HALF"
objs=""
for f in *.hip; do
  b=${f%.hip}
  if grep -q "tri_dev.h" $f; then
    extra=""; case $b in knn|knn_stream|knn_duo) extra="-fno-honor-nans";; esac
    hipcc $F $extra -c $f -o /tmp/trihalf/$b.o &
    objs="$objs /tmp/trihalf/$b.o"
  else
    objs="$objs build/$b.o"
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/lib_tri_half.so $objs
