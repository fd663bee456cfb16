#!/bin/bash
# device blocks aligned to 128 bytes (stride 2752 -> 2816 at D = 128 / 4-bit, 1728 -> 1792 at 2-bit) against 64
export TMPDIR=/tmp
O=gpurun_out/r3b_step14; mkdir -p $O
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || exit 1
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 3 product build/libcph_a128.so | tee $O/ab_c2.txt
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 3 --nq 100000 product build/libcph_a128.so | tee $O/ab_c2_100k.txt
python3 bench.py --config recall --steps 2 --warmup 1 --no-cpu-baseline > $O/prep_r.json 2> $O/prep_r.err || exit 1
python3 scripts/ab_libs.py --config recall --k 20 --rounds 2 product build/libcph_a128.so | tee $O/ab_recall.txt
