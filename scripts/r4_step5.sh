#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4_step5; mkdir -p $O
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || { tail -5 $O/prep_c2.err; exit 1; }
for l in timers r3_timers; do
python3 scripts/phase_timers.py --lib build/libcph_$l.so --config c2 --reps 3 > $O/$l.json 2> $O/$l.err; echo $l; grep "phase cycles" $O/$l.err | tail -1; cut -c1-120 $O/$l.json
done
