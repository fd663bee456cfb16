#!/bin/bash
# final tree: GPU test tier, single-query latency (small-batch launch without probe-first), default line
export TMPDIR=/tmp
O=gpurun_out/r3b_step26; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit 1
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
python3 -c "
import json; j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); r=j['roofline']; print('default: value', round(j['value']), 'ms/step', round(j['ms_per_step'],4), 'kernel_ms', r['kernel_ms'], 'frac', round(r['frac'],4), 'traffic', round(r['traffic'],1), 'full queue', round(r['full_queue']['kernel_ms'],3), round(r['full_queue']['frac'],4), 'gate', round(j['qps_at_recall_gate']['value']), 'legs_failed', j.get('legs_failed'), 'parity', j['cpu_baseline']['parity_vs_reference'], 'cpu', round(j['cpu_baseline']['value']))"
python3 scripts/single_query_latency.py c2 | tee $O/single_query_latency.json
