#!/bin/bash
# round 4, step 1: parity suite on the new tree (short lists, exact stage-2 gate under probe first), then C2 kernel A/B
export TMPDIR=/tmp
O=gpurun_out/r4_step1; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log; [ $rc -eq 0 ] || exit 1
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || { tail -5 $O/prep_c2.err; exit 1; }
python3 scripts/ab_libs.py --config c2 --rounds 3 product build/libcph_r3.so build/libcph_launder.so | tee $O/ab_c2.txt
python3 scripts/phase_timers.py --product --config c2 --reps 3 | tee $O/stats_product.json
python3 scripts/ab_libs.py --config c2 --nq 100000 --rounds 2 product build/libcph_r3.so build/libcph_launder.so | tee $O/ab_c2_100k.txt
