#!/bin/bash
# probe first, then the codes / aux values of the NEW neighbours only (4-bit, D = 128; -DCPH_PROBE_FIRST): parity, A/B, bench line
export TMPDIR=/tmp
O=gpurun_out/r3b_step20; mkdir -p $O
CPH_LIB_PATH=$PWD/build/libcph_pf.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "search" > $O/pytest_pf.log 2>&1; rc=$?; tail -3 $O/pytest_pf.log; [ $rc -eq 0 ] || exit 1
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || exit 1
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 3 product build/libcph_pf.so | tee $O/ab_c2.txt
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 2 --nq 100000 product build/libcph_pf.so | tee $O/ab_c2_100k.txt
CPH_LIB_PATH=$PWD/build/libcph_pf.so python3 bench.py --no-extra-legs --cpu-queries 2000 > $O/bench_pf.json 2> $O/bench_pf.err; echo "bench rc=$?"
python3 -c "
import json; j=json.loads(open('$O/bench_pf.json').read().strip().splitlines()[-1]); print('probe-first: value', round(j['value']), 'ms/step', round(j['ms_per_step'],4), 'kernel_ms', j['roofline']['kernel_ms'], 'full queue ms', round(j['roofline']['full_queue']['kernel_ms'],3), 'parity', j['cpu_baseline']['parity_vs_reference'])"
python3 bench.py --no-extra-legs --no-cpu-baseline > $O/bench_prod.json 2> $O/bench_prod.err
python3 -c "
import json; j=json.loads(open('$O/bench_prod.json').read().strip().splitlines()[-1]); print('product:     value', round(j['value']), 'ms/step', round(j['ms_per_step'],4), 'kernel_ms', j['roofline']['kernel_ms'], 'full queue ms', round(j['roofline']['full_queue']['kernel_ms'],3))"
