#!/bin/bash
# kNN kernel timing (kernel-trace) for the default build and an A/B build of knn_tri.hip: tools/knn_ab.sh <tag> "<EXTRA flags>"
tag=$1; extra=$2
python -m pytest tests/test_gpu_stages.py -q -x -k "knn" > gpurun_out/${tag}_knn_tests.log 2>&1; tail -3 gpurun_out/${tag}_knn_tests.log
tools/kstats_any.sh ${tag}_a tools/knn_probe.py 2048 20 | head -4
if [ -n "$extra" ]; then
  touch samble_amd/csrc/knn_tri.hip; make -C samble_amd/csrc EXTRA="$extra" > /dev/null 2>&1
  python -m pytest tests/test_gpu_stages.py -q -x -k "knn" > gpurun_out/${tag}_knn_tests_b.log 2>&1; tail -3 gpurun_out/${tag}_knn_tests_b.log
  tools/kstats_any.sh ${tag}_b tools/knn_probe.py 2048 20 | head -4
fi
