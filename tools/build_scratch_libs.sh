#!/bin/bash
# Scratch builds of the library for the diagnostics under tools/ (never shipped, git-ignored under tools/scratch/):
#   tools/scratch/lib_stamps.so     chain.hip / select.hip with -DSAMBLE_STAMPS   (s_memtime marks, tools/tail_stamps.py)
#   tools/scratch/lib_rc_stamps.so  attn_tri.hip / attn_bwd_tri.hip with -DSAMBLE_STAMPS (tools/rc_stamps.py: the
#                                   iterations of attn_rows_rc_tri, attn_stats_nl_tri, bwd_kacc_tri; the producer /
#                                   consumer kernel as well when built with -DSAMBLE_ROWS_PRODUCER_CONSUMER)
#   tools/scratch/abl/lib_<m>.so    attn_tri.hip with -DSAMBLE_RC_ABL=<m> (timing-only ablations, tools/abl_rows_rc.py)
set -e
cd "$(dirname "$0")/../samble_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/scratch/abl
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden"
for f in chain select; do hipcc $F -ffp-contract=off -DSAMBLE_STAMPS -c $f.hip -o /tmp/${f}_st.o; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/lib_stamps.so $(ls build/*.o | grep -v "chain.o\|select.o") /tmp/chain_st.o /tmp/select_st.o
for f in attn_tri attn_bwd_tri attn_bwd proj_tri; do hipcc $F -DSAMBLE_STAMPS $EXTRA -c $f.hip -o /tmp/${f}_st.o; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/lib_rc_stamps.so $(ls build/*.o | grep -v "attn_tri.o\|attn_bwd_tri.o\|attn_bwd.o\|proj_tri.o") /tmp/attn_tri_st.o /tmp/attn_bwd_tri_st.o /tmp/attn_bwd_st.o /tmp/proj_tri_st.o
for a in 0 1 2 4 6; do
  hipcc $F -DSAMBLE_RC_ABL=$a -c attn_tri.hip -o /tmp/attn_tri_$a.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/abl/lib_$a.so $(ls build/*.o | grep -v attn_tri.o) /tmp/attn_tri_$a.o
done
# tools/scratch/nlabl/lib_<name>.so: attn_stats_nl_tri timing-only ablations (tools/abl_stats_nl.py)
mkdir -p ../../tools/scratch/nlabl
nl() {  # name, flags...
  local name=$1; shift
  hipcc $F "$@" -c attn_tri.hip -o /tmp/attn_tri_nl_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/nlabl/lib_$name.so $(ls build/*.o | grep -v attn_tri.o) /tmp/attn_tri_nl_$name.o
}
nl nomfma -DSAMBLE_NL_ABL=2
nl halfreads -DSAMBLE_NL_ABL=16
nl nobar -DSAMBLE_NL_NOBAR
nl nobar_nomfma -DSAMBLE_NL_NOBAR -DSAMBLE_NL_ABL=2
nl noextract -DSAMBLE_NL_ABL=32
nl noexp -DSAMBLE_NL_ABL=64
