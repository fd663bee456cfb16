#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4_pfnarrow; mkdir -p $O
python3 bench.py --config recall1m --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs --recall-queries 20 > $O/prep.json 2> $O/prep.err || { tail -5 $O/prep.err; exit 1; }
for r in 1 2; do for l in product build/libcph_pfnarrow.so; do
if [ $l = product ]; then A="--product"; else A="--lib $l"; fi
python3 scripts/phase_timers.py $A --config recall1m --k 20 --reps 2 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['lib'],'best_kernel_us',j['best_kernel_us'])"
done; done | tee $O/summary.txt
