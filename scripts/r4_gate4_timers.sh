#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4_gate4_timers; mkdir -p $O
python3 bench.py --config recall1m --bits 4 --k 500 --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs --recall-queries 50 > $O/prep.json 2> $O/prep.err || { tail -5 $O/prep.err; exit 1; }
for l in timers fine; do
python3 scripts/phase_timers.py --lib build/libcph_$l.so --config recall1m --bits 4 --k 500 --reps 2 > $O/$l.json 2> $O/$l.err; echo $l; grep "cycles" $O/$l.err | tail -1; cut -c1-400 $O/$l.json
done
python3 scripts/phase_timers.py --lib build/libcph_timers.so --config recall1m --bits 4 --k 100 --reps 2 > $O/t100.json 2> $O/t100.err; echo k100; grep "cycles" $O/t100.err | tail -1; cut -c1-400 $O/t100.json
