#!/bin/bash
# an LDS filter of the ids a query has estimated: codes of neighbours it clears go out with the probe (-DCPH_BLOOM)
export TMPDIR=/tmp
O=gpurun_out/r3b_step28; mkdir -p $O
CPH_LIB_PATH=$PWD/build/libcph_bloom.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "search" > $O/pytest_bloom.log 2>&1; rc=$?; tail -3 $O/pytest_bloom.log; [ $rc -eq 0 ] || exit 1
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || exit 1
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 3 product build/libcph_bloom.so | tee $O/ab_c2.txt
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 2 --nq 100000 product build/libcph_bloom.so | tee $O/ab_c2_100k.txt
CPH_LIB_PATH=$PWD/build/libcph_bloom.so python3 bench.py --no-extra-legs --cpu-queries 2000 > $O/bench_bloom.json 2> $O/bench_bloom.err
python3 -c "
import json; j=json.loads(open('$O/bench_bloom.json').read().strip().splitlines()[-1]); print('bloom: value', round(j['value']), 'ms/step', round(j['ms_per_step'],4), 'kernel_ms', j['roofline']['kernel_ms'], 'full queue ms', round(j['roofline']['full_queue']['kernel_ms'],3), 'parity', j['cpu_baseline']['parity_vs_reference'])"
python3 bench.py --no-extra-legs --no-cpu-baseline > $O/bench_prod.json 2> $O/bench_prod.err
python3 -c "
import json; j=json.loads(open('$O/bench_prod.json').read().strip().splitlines()[-1]); print('product: value', round(j['value']), 'ms/step', round(j['ms_per_step'],4), 'kernel_ms', j['roofline']['kernel_ms'], 'full queue ms', round(j['roofline']['full_queue']['kernel_ms'],3))"
