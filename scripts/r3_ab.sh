#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r3_ab; mkdir -p $O
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err
python3 bench.py --config recall --steps 2 --warmup 1 --no-cpu-baseline > $O/prep_r.json 2> $O/prep_r.err
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 3 product build/libcph_r2.so build/libcph_r3a.so | tee $O/ab_c2.txt
python3 scripts/ab_libs.py --config recall --k 20 --rounds 2 product build/libcph_r2.so build/libcph_r3a.so | tee $O/ab_recall.txt
