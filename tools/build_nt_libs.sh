#!/bin/bash
# Scratch libraries for the map-traffic cache-policy A/B (tools/ab_lib.sh): tools/scratch/nt/lib_<mask>.so =
# the library with -DSAMBLE_MAP_NT=<mask> (csrc/tri_dev.h)
set -e
cd "$(dirname "$0")/../samble_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/scratch/nt
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden"
for m in ${MASKS:-0 1 2 4 8 15}; do
  hipcc $F -DSAMBLE_MAP_NT=$m -c attn_tri.hip -o /tmp/attn_tri_nt$m.o &
  hipcc $F -DSAMBLE_MAP_NT=$m -c attn_bwd_tri.hip -o /tmp/attn_bwd_tri_nt$m.o &
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/nt/lib_$m.so $(ls build/*.o | grep -v "attn_tri.o\|attn_bwd_tri.o") /tmp/attn_tri_nt$m.o /tmp/attn_bwd_tri_nt$m.o
done
