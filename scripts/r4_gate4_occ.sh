#!/bin/bash
# 4-bit gate point (k = 500): how much does the kernel gain per resident wave?  (LDS allows 17 per CU at this k)
export TMPDIR=/tmp
O=gpurun_out/r4_gate4_occ; mkdir -p $O
python3 bench.py --config recall1m --bits 4 --k 500 --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs --recall-queries 20 > $O/prep.json 2> $O/prep.err || { tail -5 $O/prep.err; exit 1; }
for W in 9 13 17; do
CPH_WAVES_PER_CU=$W python3 scripts/phase_timers.py --product --config recall1m --bits 4 --k 500 --reps 3 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves/CU $W', 'best_kernel_us',j['best_kernel_us'],'slots',j['stats']['slots'])"
done | tee $O/summary.txt
for W in 16 20 24; do
CPH_WAVES_PER_CU=$W python3 scripts/phase_timers.py --product --config recall1m --bits 4 --k 200 --reps 3 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k=200 waves/CU $W', 'best_kernel_us',j['best_kernel_us'],'slots',j['stats']['slots'])"
done | tee -a $O/summary.txt
