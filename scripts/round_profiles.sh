#!/bin/bash
# GPU suite, concurrent-caller throughput (ours and the reference), and the rocprofv3 kernel trace of the C2 bench command
export TMPDIR=/tmp
O=${1:-gpurun_out/round_profiles}; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit 1
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || { tail -5 $O/prep_c2.err; exit 1; }
python3 scripts/concurrent_search.py c2 | tee $O/concurrent_c2.json
python3 scripts/single_query_latency.py c2 | tee $O/single_query_latency.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-extra-legs --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_traced_c2.json 2> $GRAFT_REPO_ROOT/$O/bench_traced_c2.err
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); echo "trace file: $f"
[ -n "$f" ] && python3 scripts/kernel_trace_summary.py $f --serial 20 | tee $O/kernel_trace_summary.txt
g=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$g" ] && cp $g $O/kernel_stats.csv
rm -rf $O/trace
python3 -c "
import json; j=json.loads(open('$O/bench_traced_c2.json').read().strip().splitlines()[-1]); print('traced run: kernel_ms (HIP events)', j['roofline']['kernel_ms'], 'value', round(j['value']))"
