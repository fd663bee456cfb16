#!/bin/bash
# next-top prefetch on top of probe-first (-DCPH_PF_PREFETCH): parity, A/B, bench lines
export TMPDIR=/tmp
O=gpurun_out/r3b_step25; mkdir -p $O
CPH_LIB_PATH=$PWD/build/libcph_pfp.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "search" > $O/pytest_pfp.log 2>&1; rc=$?; tail -3 $O/pytest_pfp.log; [ $rc -eq 0 ] || exit 1
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || exit 1
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 3 product build/libcph_pfp.so | tee $O/ab_c2.txt
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 2 --nq 100000 product build/libcph_pfp.so | tee $O/ab_c2_100k.txt
for l in pfp; do
CPH_LIB_PATH=$PWD/build/libcph_$l.so python3 bench.py --no-extra-legs --cpu-queries 2000 > $O/bench_$l.json 2> $O/bench_$l.err
python3 -c "
import json; j=json.loads(open('$O/bench_$l.json').read().strip().splitlines()[-1]); print('$l: value', round(j['value']), 'ms/step', round(j['ms_per_step'],4), 'kernel_ms', j['roofline']['kernel_ms'], 'full queue ms', round(j['roofline']['full_queue']['kernel_ms'],3), 'parity', j['cpu_baseline']['parity_vs_reference'])"
done
python3 bench.py --no-extra-legs --no-cpu-baseline > $O/bench_prod.json 2> $O/bench_prod.err
python3 -c "
import json; j=json.loads(open('$O/bench_prod.json').read().strip().splitlines()[-1]); print('product: value', round(j['value']), 'ms/step', round(j['ms_per_step'],4), 'kernel_ms', j['roofline']['kernel_ms'], 'full queue ms', round(j['roofline']['full_queue']['kernel_ms'],3))"
CPH_LIB_PATH=$PWD/build/libcph_pfp.so python3 scripts/single_query_latency.py c2 | tee $O/single_pfp.json
