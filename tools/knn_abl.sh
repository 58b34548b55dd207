#!/bin/bash
for abl in 0 1 2 4 8 3 11 15; do
  echo "== ABL=$abl $1"
  bash tools/knn_stamp_run.sh "-DSAMBLE_KNN_ABL=$abl $1" 2>/dev/null | grep -E "seed|prod|barrier|total"
done
