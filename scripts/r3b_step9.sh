#!/bin/bash
# symmetric kNN self-join: exactness tests, stage times at 1M x 128 / 262k x 128 against the two-pass kernel, builder tests
export TMPDIR=/tmp
O=gpurun_out/r3b_step9; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_builder.py -x -q -m gpu -k "knn" > $O/pytest_knn.log 2>&1; rc=$?; tail -15 $O/pytest_knn.log; [ $rc -eq 0 ] || exit 1
CPH_BUILD_VERBOSE=1 timeout -k 10 300 python3 scripts/knn_time.py 262144 128 2>&1 | tee $O/knn_sym_262k.txt
CPH_KNN_SYM=0 timeout -k 10 300 python3 scripts/knn_time.py 262144 128 2>&1 | tee $O/knn_old_262k.txt
CPH_BUILD_VERBOSE=1 timeout -k 10 300 python3 scripts/knn_time.py 1000000 128 2>&1 | tee $O/knn_sym_1m.txt
CPH_KNN_SYM=0 timeout -k 10 300 python3 scripts/knn_time.py 1000000 128 2>&1 | tee $O/knn_old_1m.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_builder.py -x -q -m gpu > $O/pytest_builder.log 2>&1; tail -5 $O/pytest_builder.log
