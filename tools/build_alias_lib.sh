#!/bin/bash
# Scratch library for the on-die-maps upper bound (tools/experiments/map_alias.md): tools/scratch/alias/lib_alias.so =
# the library with -DSAMBLE_MAP_ALIAS (csrc/samble_dev.h map_cloud): timing only, results are wrong.
set -e
cd "$(dirname "$0")/../samble_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/scratch/alias
for mode in ${MODES:-1 2 3 4}; do
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DSAMBLE_MAP_ALIAS=$mode"
  hipcc $F -c attn_tri.hip -o /tmp/attn_tri_alias.o &
  hipcc $F -c attn_bwd_tri.hip -o /tmp/attn_bwd_tri_alias.o &
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/alias/lib_alias$mode.so $(ls build/*.o | grep -v "attn_tri.o\|attn_bwd_tri.o") /tmp/attn_tri_alias.o /tmp/attn_bwd_tri_alias.o
done
cp ../libsamble_hip.so ../../tools/scratch/alias/lib_ship.so
ls -la ../../tools/scratch/alias/
