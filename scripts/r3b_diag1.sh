#!/bin/bash
# fine phase timers (CPH_PHASE_TIMERS=2) and spilled-beam traffic counts on the two gate workloads
export TMPDIR=/tmp
O=gpurun_out/r3b_diag1; mkdir -p $O
python3 bench.py --config recall --steps 2 --warmup 1 --no-cpu-baseline > $O/prep_r.json 2> $O/prep_r.err || exit 1
python3 scripts/phase_timers.py --config recall --k 20 --lib build/libcph_fine.so > $O/fine_recall.json 2> $O/fine_recall.err || exit 1
grep "fine cycles" $O/fine_recall.err | tail -1
python3 bench.py --config recall1m --steps 2 --warmup 1 --no-cpu-baseline > $O/prep_r1m.json 2> $O/prep_r1m.err || exit 1
python3 scripts/phase_timers.py --config recall1m --k 20 --reps 2 --lib build/libcph_fine.so > $O/fine_recall1m.json 2> $O/fine_recall1m.err || exit 1
grep "fine cycles" $O/fine_recall1m.err | tail -1
python3 scripts/phase_timers.py --config recall1m --k 20 --reps 1 --lib build/libcph_traffic.so > $O/traffic_recall1m.json 2> $O/traffic_recall1m.err || exit 1
grep "traffic" $O/traffic_recall1m.err | tail -1
python3 scripts/phase_timers.py --config recall1m --k 20 --reps 2 --product > $O/prod_recall1m.json 2> $O/prod_recall1m.err || exit 1
cat $O/prod_recall1m.json
