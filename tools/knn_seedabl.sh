#!/bin/bash
tools/knn_stamp_run.sh "" 2048 2>&1 | tail -13
