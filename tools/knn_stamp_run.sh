#!/bin/bash
touch samble_amd/csrc/knn_tri.hip; make -C samble_amd/csrc EXTRA="-DSAMBLE_KNN_STAMP $1" > /dev/null 2>&1
python tools/knn_stamps.py
