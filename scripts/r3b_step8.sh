#!/bin/bash
# why the prefetch wave does not pay: fine timers of a 32-query launch with and without it
export TMPDIR=/tmp
O=gpurun_out/r3b_step8; mkdir -p $O
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || exit 1
for nq in 1 32; do
python3 scripts/phase_timers.py --config c2 --k 10 --nq $nq --reps 5 --lib build/libcph_fine.so > $O/fine_help_$nq.json 2> $O/fine_help_$nq.err
echo "help nq=$nq: $(grep 'fine cycles' $O/fine_help_$nq.err | tail -1)"; cat $O/fine_help_$nq.json | cut -c1-300
CPH_NO_HELPER_WAVE=1 python3 scripts/phase_timers.py --config c2 --k 10 --nq $nq --reps 5 --lib build/libcph_fine.so > $O/fine_nohelp_$nq.json 2> $O/fine_nohelp_$nq.err
echo "nohelp nq=$nq: $(grep 'fine cycles' $O/fine_nohelp_$nq.err | tail -1)"; cat $O/fine_nohelp_$nq.json | cut -c1-300
done
