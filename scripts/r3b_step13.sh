#!/bin/bash
# final tree of the round: GPU test tier, rocprofv3 trace of the C2 part of the bench, the default line, kNN / build times
export TMPDIR=/tmp
O=gpurun_out/r3b_step13; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit 1
CPH_BUILD_VERBOSE=1 timeout -k 10 300 python3 scripts/knn_time.py 1000000 128 2>&1 | tee $O/knn_sym_1m.txt
CPH_BUILD_VERBOSE=1 timeout -k 10 600 python3 scripts/time_build.py 2>&1 | grep -E "build\]|build\+finalize|recall" | tee $O/time_build_c2.txt
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra-legs --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_traced.json 2> $GRAFT_REPO_ROOT/$O/bench_traced.err; cd $GRAFT_REPO_ROOT
python3 scripts/kernel_trace_summary.py $(ls $O/trace/*/*kernel_trace.csv $O/trace/*kernel_trace.csv 2>/dev/null | head -1) --serial 23 > $O/kernel_trace_summary.txt 2>&1; head -12 $O/kernel_trace_summary.txt; tail -2 $O/kernel_trace_summary.txt
cp $(ls $O/trace/*/*kernel_stats.csv $O/trace/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/trace
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
tail -c 600 $O/bench_default.json; echo
