#!/bin/bash
# Build the rows-pair experiment beside the library's objects and run it: tools/experiments/run_rows_pair.sh ["<ablation masks>"]
#   tools/scratch/pair/lib_<mask>.so   (mask 0 = the kernel as it is; others: timing only, see SAMBLE_PR_ABL), all
#   with -DSAMBLE_STAMPS (s_memtime marks of one pair of waves, one tile)
set -e
here="$(cd "$(dirname "$0")" && pwd)"
cd "$here/../../samble_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/scratch/pair; rm -f ../../tools/scratch/pair/*.so
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -w"
for a in ${1:-0}; do
  hipcc $F -DSAMBLE_STAMPS -DSAMBLE_PR_ABL=$a -c "$here/attn_rows_pair.hip" -o /tmp/attn_rows_pair_$a.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/pair/lib_$a.so build/*.o /tmp/attn_rows_pair_$a.o
done
cd "$here/../.."
[ -n "$NO_RUN" ] || python3 tools/experiments/rows_pair.py
