#!/bin/bash
# a query that turns out long (> N expansions) raises its wave's priority: isolated batch, kernel alone
export TMPDIR=/tmp
O=gpurun_out/r3b_step19; mkdir -p $O
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || exit 1
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 3 product build/libcph_prio200.so build/libcph_prio300.so build/libcph_prio400.so | tee $O/ab_c2.txt
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 2 --nq 100000 product build/libcph_prio300.so | tee $O/ab_c2_100k.txt
