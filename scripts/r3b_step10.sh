#!/bin/bash
# symmetric kNN with batched append atomics: tests + stage times (1M x 128, 1M x 960 -> D = 1024 at 300k rows), build times
export TMPDIR=/tmp
O=gpurun_out/r3b_step10; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_builder.py -x -q -m gpu -k "knn" > $O/pytest_knn.log 2>&1; rc=$?; tail -3 $O/pytest_knn.log; [ $rc -eq 0 ] || exit 1
CPH_BUILD_VERBOSE=1 timeout -k 10 300 python3 scripts/knn_time.py 1000000 128 2>&1 | tee $O/knn_sym_1m.txt
CPH_KNN_SYM=0 timeout -k 10 300 python3 scripts/knn_time.py 1000000 128 2>&1 | tee $O/knn_old_1m.txt
CPH_BUILD_VERBOSE=1 timeout -k 10 300 python3 scripts/knn_time.py 300000 960 2>&1 | tee $O/knn_sym_300k_960.txt
CPH_KNN_SYM=0 timeout -k 10 300 python3 scripts/knn_time.py 300000 960 2>&1 | tee $O/knn_old_300k_960.txt
CPH_BUILD_VERBOSE=1 timeout -k 10 600 python3 scripts/time_build.py 2>&1 | tail -25 | tee $O/time_build.txt
