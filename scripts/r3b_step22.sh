#!/bin/bash
# probe-first as the default of search_kernel<4,128>: GPU test tier, PMC passes on C2, C4 A/B against -DCPH_NO_PROBE_FIRST, default line
export TMPDIR=/tmp
O=gpurun_out/r3b_step22; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit 1
bash scripts/pmc_search.sh $O/pmc_c2 c2 10 > $O/pmc_c2.log 2>&1; tail -22 $O/pmc_c2/summary.txt
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
python3 -c "
import json; j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); r=j['roofline']; print('default: value', round(j['value']), 'ms/step', round(j['ms_per_step'],4), 'kernel_ms', r['kernel_ms'], 'frac', round(r['frac'],4), 'traffic', r['traffic'], 'full queue', {k:(round(v,4) if isinstance(v,float) else v) for k,v in r['full_queue'].items() if k!='measured'}, 'gate', round(j['qps_at_recall_gate']['value']), 'legs_failed', j.get('legs_failed'), 'parity', j['cpu_baseline']['parity_vs_reference'])"
python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $O/prep_c4.json 2> $O/prep_c4.err || exit 1
python3 scripts/ab_libs.py --config c4 --k 10 --rounds 2 product build/libcph_nopf.so | tee $O/ab_c4.txt
python3 scripts/ab_libs.py --config c4 --k 10 --rounds 2 --nq 100000 product build/libcph_nopf.so | tee $O/ab_c4_100k.txt
