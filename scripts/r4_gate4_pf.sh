#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4_gate4_pf; mkdir -p $O
python3 bench.py --config recall1m --bits 4 --k 500 --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs --recall-queries 50 > $O/prep.json 2> $O/prep.err || { tail -5 $O/prep.err; exit 1; }
for K in 500 100; do for l in base nopf; do
python3 scripts/phase_timers.py --lib build/libcph_$l.so --config recall1m --bits 4 --k $K --reps 3 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k',j['k'],j['lib'],'best_kernel_us',j['best_kernel_us'],'slots',j['stats']['slots'])"
done; done | tee $O/summary.txt
