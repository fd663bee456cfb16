#!/bin/bash
# access-shape ceiling of the gate workload, PMC passes of the final kernel (recall, c2), rocprofv3 trace of the c2 bench, the default line
export TMPDIR=/tmp
O=gpurun_out/r3b_step6; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 scripts/micro/gate_shape.hip -o /tmp/gate_shape 2> /dev/null || exit 1
/tmp/gate_shape 100000 | tee $O/gate_shape_100k.txt
/tmp/gate_shape 1000000 24 | tee $O/gate_shape_1m.txt
bash scripts/pmc_search.sh $O/pmc_recall recall 20 > $O/pmc_recall.log 2>&1; tail -25 $O/pmc_recall/summary.txt
bash scripts/pmc_search.sh $O/pmc_c2 c2 10 > $O/pmc_c2.log 2>&1; tail -25 $O/pmc_c2/summary.txt
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra-legs --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_traced.json 2> $GRAFT_REPO_ROOT/$O/bench_traced.err; cd $GRAFT_REPO_ROOT
python3 scripts/kernel_trace_summary.py $(ls $O/trace/*/*kernel_trace.csv $O/trace/*kernel_trace.csv 2>/dev/null | head -1) --serial 23 > $O/kernel_trace_summary.txt 2>&1; cat $O/kernel_trace_summary.txt | head -12; tail -3 $O/kernel_trace_summary.txt
cp $(ls $O/trace/*/*kernel_stats.csv $O/trace/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/trace
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
tail -c 2500 $O/bench_default.json
