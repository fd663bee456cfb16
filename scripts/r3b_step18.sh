#!/bin/bash
# express CUs for the head of the launch order: isolated 10,000-query batch, kernel alone
export TMPDIR=/tmp
O=gpurun_out/r3b_step18; mkdir -p $O
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || exit 1
run() { echo "express=$1 keep=$2: 10k $(CPH_EXPRESS=$1 CPH_EXPRESS_KEEP=$2 python3 scripts/phase_timers.py --product --config c2 --k 10 --nq 10000 --reps 9 2>/dev/null | python3 -c 'import sys,json; print(json.loads(sys.stdin.read().splitlines()[-1])["best_kernel_us"])') us" | tee -a $O/express_sweep.txt; }
run 0 4; run 100 4; run 200 4; run 400 4; run 200 8; run 400 8; run 800 8; run 0 4; run 200 2; run 1000 12
CPH_EXPRESS=200 python3 bench.py --config c2 --steps 10 --warmup 2 --no-extra-legs --cpu-queries 2000 > $O/bench_express.json 2> $O/bench_express.err; echo "bench rc=$?"
python3 -c "
import json; j=json.loads(open('$O/bench_express.json').read().strip().splitlines()[-1]); print('value', round(j['value']), 'kernel_ms', j['roofline']['kernel_ms'], 'frac', round(j['roofline']['frac'],4), 'qps_serial', round(j['qps_serial']), 'parity', j['cpu_baseline']['parity_vs_reference'])"
