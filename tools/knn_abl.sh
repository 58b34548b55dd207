#!/bin/bash
for abl in 0 1 2 4 7; do
  echo "== SEEDABL=$abl"
  bash tools/knn_stamp_run.sh "-DSAMBLE_KNN_SEEDABL=$abl" 2>/dev/null | grep -E "^seed"
done
