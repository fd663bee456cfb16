#!/bin/bash
# kernel alone against resident waves per CU (latency of the longest query vs throughput)
export TMPDIR=/tmp
O=gpurun_out/r3b_step16; mkdir -p $O
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || exit 1
for w in 24 22 20 18 16 12; do
  echo "waves per CU $w: 10k $(CPH_WAVES_PER_CU=$w python3 scripts/phase_timers.py --product --config c2 --k 10 --nq 10000 --reps 7 2>/dev/null | python3 -c 'import sys,json; print(json.loads(sys.stdin.read().splitlines()[-1])["best_kernel_us"])') us, 100k $(CPH_WAVES_PER_CU=$w python3 scripts/phase_timers.py --product --config c2 --k 10 --nq 100000 --reps 4 2>/dev/null | python3 -c 'import sys,json; print(json.loads(sys.stdin.read().splitlines()[-1])["best_kernel_us"])') us" | tee -a $O/waves_sweep.txt
done
