#!/bin/bash
# bisect of the +5 % against HEAD: which of the pieces costs what (C2, gate workloads)
export TMPDIR=/tmp
O=gpurun_out/r3b_step4; mkdir -p $O
B=build
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || exit 1
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 3 product $B/libcph_base.so $B/libcph_NOANC.so $B/libcph_OLD_POP.so $B/libcph_NOSKIP.so $B/libcph_OLDPOP_NOANC.so $B/libcph_lh.so | tee $O/ab_c2.txt
python3 bench.py --config recall --steps 2 --warmup 1 --no-cpu-baseline > $O/prep_r.json 2> $O/prep_r.err || exit 1
python3 scripts/ab_libs.py --config recall --k 20 --rounds 2 product $B/libcph_base.so $B/libcph_NOANC.so $B/libcph_OLD_POP.so $B/libcph_NOSKIP.so $B/libcph_OLDPOP_NOANC.so $B/libcph_lh.so $B/libcph_w7.so | tee $O/ab_recall.txt
python3 bench.py --config recall1m --steps 2 --warmup 1 --no-cpu-baseline > $O/prep_r1m.json 2> $O/prep_r1m.err || exit 1
python3 scripts/ab_libs.py --config recall1m --k 20 --rounds 2 product $B/libcph_base.so $B/libcph_OLD_POP.so $B/libcph_w7.so | tee $O/ab_recall1m.txt
