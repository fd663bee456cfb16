#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_ab3; mkdir -p $O
python3 bench.py --config recall --steps 2 --warmup 1 --no-cpu-baseline > $O/prep_r.json 2> $O/prep_r.err
python3 scripts/ab_libs.py --config recall --k 20 --rounds 2 product build/libcph_w7.so | tee $O/ab_recall_w7.txt
python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $O/prep_c4.json 2> $O/prep_c4.err; echo "c4 prep rc=$?"
python3 scripts/ab_libs.py --config c4 --k 10 --rounds 2 product build/libcph_r2.so | tee $O/ab_c4.txt
python3 scripts/ab_libs.py --config c4 --k 10 --rounds 2 --nq 100000 product build/libcph_r2.so | tee $O/ab_c4_100k.txt
python3 scripts/phase_timers.py --config c4 --k 10 --lib build/libcph_traffic.so 2>&1 | grep traffic | tail -1
