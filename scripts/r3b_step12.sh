#!/bin/bash
# symmetric join with the tile epilogue pipelined behind the next tile's MFMAs (two accumulator sets): tests, A/B against the plain join
export TMPDIR=/tmp
O=gpurun_out/r3b_step12; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_builder.py -x -q -m gpu -k "knn" > $O/pytest_knn.log 2>&1; rc=$?; tail -3 $O/pytest_knn.log; [ $rc -eq 0 ] || exit 1
for i in 1 2; do
CPH_BUILD_VERBOSE=1 timeout -k 10 300 python3 scripts/knn_time.py 1000000 128 2>&1 | grep -E "join|wall" | tee -a $O/knn_pipe_1m.txt
CPH_KNN_SYM_NOPIPE=1 CPH_BUILD_VERBOSE=1 timeout -k 10 300 python3 scripts/knn_time.py 1000000 128 2>&1 | grep -E "join|wall" | tee -a $O/knn_nopipe_1m.txt
done
CPH_BUILD_VERBOSE=1 timeout -k 10 300 python3 scripts/knn_time.py 300000 960 2>&1 | grep -E "join|wall" | tee -a $O/knn_pipe_300k_960.txt
CPH_KNN_SYM_NOPIPE=1 CPH_BUILD_VERBOSE=1 timeout -k 10 300 python3 scripts/knn_time.py 300000 960 2>&1 | grep -E "join|wall" | tee -a $O/knn_nopipe_300k_960.txt
