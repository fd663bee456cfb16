#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3b_step17; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_builder.py -x -q -m gpu -k "knn" > $O/pytest_knn.log 2>&1; rc=$?; tail -12 $O/pytest_knn.log; [ $rc -eq 0 ] || exit 1
