#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r3_diag2; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1 || true
python3 bench.py --config recall --steps 2 --warmup 1 --no-cpu-baseline --no-recall-leg > $O/prep.json 2> $O/prep.err
for w in 8 12 16 20 24; do
  CPH_WAVES_PER_CU=$w python3 scripts/phase_timers.py --product --config recall --k 20 2>/dev/null | tail -1 | sed "s/^/waves_per_cu=$w /" >> $O/waves.log
done
python3 scripts/phase_timers.py --config recall --k 20 --lib build/libcph_nowait.so 2>/dev/null | tail -1 | sed "s/^/nowait /" >> $O/waves.log
cat $O/waves.log
