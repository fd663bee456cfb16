#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4_ab3; mkdir -p $O
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || { tail -5 $O/prep_c2.err; exit 1; }
python3 scripts/ab_libs.py --config c2 --rounds 3 product build/libcph_base.so build/libcph_r3.so | tee $O/ab_c2.txt
python3 scripts/ab_libs.py --config c2 --nq 100000 --rounds 2 product build/libcph_base.so build/libcph_r3.so | tee $O/ab_c2_100k.txt
