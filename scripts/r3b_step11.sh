#!/bin/bash
# final tree: GPU test tier, default bench line, C4 line (10M x 96: build with the symmetric join), MFMA counters of the join
export TMPDIR=/tmp
O=gpurun_out/r3b_step11; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit 1
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
tail -c 1200 $O/bench_default.json; echo
bash scripts/prof_knn.sh $O/prof_knn_sym 262144 128 knn_sym > $O/prof_knn_sym.log 2>&1; tail -30 $O/prof_knn_sym/summary.txt
CPH_BUILD_VERBOSE=1 python3 bench.py --config c4 --steps 20 --warmup 3 --cpu-queries 500 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?"
grep "\[build\]" $O/bench_c4.err | tail -20
tail -c 1500 $O/bench_c4.json
