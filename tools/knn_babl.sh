#!/bin/bash
# pass-B ablations of the kNN kernel (stamped scratch builds; results are garbage, timing only)
for a in 0 1; do echo "== BABL=$a"; tools/knn_stamp_run.sh "-DSAMBLE_KNN_BABL=$a" 2048 2>&1 | grep -E "^seed |^products|^barrier|^loop_end|^total"; done
