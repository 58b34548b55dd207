#!/bin/bash
# stamped scratch build of the kNN kernel (never touches the shipped library): tools/knn_stamp_run.sh ["extra flags"] [N]
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
cd "$here/samble_amd/csrc"
mkdir -p "$here/tools/scratch"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-honor-nans -DSAMBLE_KNN_STAMP $1 -c knn_duo.hip -o /tmp/knn_duo_st.o
hipcc --offload-arch=gfx950 -shared -fPIC -o "$here/tools/scratch/lib_knn_stamps.so" $(ls build/*.o | grep -v knn_duo.o) /tmp/knn_duo_st.o
cd "$here"
python3 tools/knn_stamps.py tools/scratch/lib_knn_stamps.so ${2:-2048}
