#!/bin/bash
# kNN tests + kernel timing (kernel-trace) of the shipped build, then the stamped scratch build: tools/knn_ab.sh <tag>
tag=$1
python -m pytest tests/test_gpu_stages.py -q -x -k "knn" > gpurun_out/${tag}_knn_tests.log 2>&1; tail -3 gpurun_out/${tag}_knn_tests.log
tools/kstats_any.sh ${tag}_a tools/knn_probe.py 2048 20 | head -6
tools/knn_stamp_run.sh "" 2048 2>&1 | tail -9
