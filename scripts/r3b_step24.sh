#!/bin/bash
# final tree (probe-first): rocprofv3 trace of the C2 part of the bench, single-query latency, C4 line
export TMPDIR=/tmp
O=gpurun_out/r3b_step24; mkdir -p $O
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra-legs --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_traced.json 2> $GRAFT_REPO_ROOT/$O/bench_traced.err; cd $GRAFT_REPO_ROOT
python3 scripts/kernel_trace_summary.py $(ls $O/trace/*/*kernel_trace.csv $O/trace/*kernel_trace.csv 2>/dev/null | head -1) --serial 23 > $O/kernel_trace_summary.txt 2>&1; head -8 $O/kernel_trace_summary.txt; tail -2 $O/kernel_trace_summary.txt
cp $(ls $O/trace/*/*kernel_stats.csv $O/trace/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/trace
python3 -c "
import json; j=json.loads(open('$O/bench_traced.json').read().strip().splitlines()[-1]); print('traced run: kernel_ms', j['roofline']['kernel_ms'], 'value', round(j['value']))"
python3 scripts/single_query_latency.py c2 | tee $O/single_query_latency.json
python3 bench.py --config c4 --steps 20 --warmup 3 --cpu-queries 500 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?"
python3 -c "
import json; c=json.loads(open('$O/bench_c4.json').read().strip().splitlines()[-1]); r=c['roofline']; print('c4: value',round(c['value']),'ms/step',round(c['ms_per_step'],4),'kernel_ms',r['kernel_ms'],'frac',round(r['frac'],4),'pipelined',round(r['pipelined_frac'],4),'full queue',round(r['full_queue']['kernel_ms'],2), round(r['full_queue']['frac'],4),'build',c['config']['index_build_s'],'sets',c['config']['batch_sets'], 'parity', c['cpu_baseline'].get('parity_vs_reference'))"
