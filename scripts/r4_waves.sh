#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4_waves; mkdir -p $O
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || { tail -5 $O/prep_c2.err; exit 1; }
for W in 24 22 20 16; do for NQ in 10000 100000; do
CPH_WAVES_PER_CU=$W python3 scripts/phase_timers.py --product --config c2 --nq $NQ --reps 4 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves/CU $W nq $NQ', 'best_kernel_us',j['best_kernel_us'],'slots',j['stats']['slots'])"
done; done | tee $O/summary.txt
