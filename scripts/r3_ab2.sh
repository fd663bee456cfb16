#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r3_ab2; mkdir -p $O
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 3 --nq 100000 product build/libcph_r2.so | tee $O/ab_c2_100k.txt
python3 bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/c2.json 2> $O/c2.err
python3 -c "
import json; j=json.loads(open('$O/c2.json').read().strip().splitlines()[-1]); print('c2 qps', j['value'], 'kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'], 'serial', j['qps_serial'], 'pipelined_frac', j['roofline']['pipelined_frac'])"
CPH_LIB_PATH=build/libcph_r2.so python3 bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/c2_r2.json 2> $O/c2_r2.err
python3 -c "
import json; j=json.loads(open('$O/c2_r2.json').read().strip().splitlines()[-1]); print('c2 r2lib qps', j['value'], 'kernel_ms', j['roofline']['kernel_ms'], 'serial', j['qps_serial'])"
