#!/bin/bash
# HEAD's heap routines + laundered lane ids + single-push shortcut: tests, A/B against HEAD, scheduler strategies
export TMPDIR=/tmp
O=gpurun_out/r3b_step5; mkdir -p $O
B=build
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit 1
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || exit 1
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 3 product $B/libcph_base.so $B/libcph_ns.so $B/libcph_maxilp.so $B/libcph_maxmem.so | tee $O/ab_c2.txt
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 2 --nq 100000 product $B/libcph_base.so $B/libcph_maxilp.so $B/libcph_maxmem.so | tee $O/ab_c2_100k.txt
python3 bench.py --config recall --steps 2 --warmup 1 --no-cpu-baseline > $O/prep_r.json 2> $O/prep_r.err || exit 1
python3 scripts/ab_libs.py --config recall --k 20 --rounds 2 product $B/libcph_base.so $B/libcph_ns.so $B/libcph_maxilp.so $B/libcph_maxmem.so | tee $O/ab_recall.txt
python3 scripts/phase_timers.py --config recall --k 20 --lib build/libcph_fine.so > $O/fine_recall.json 2> $O/fine_recall.err || exit 1
grep "fine cycles" $O/fine_recall.err | tail -1
python3 bench.py --config recall1m --steps 2 --warmup 1 --no-cpu-baseline > $O/prep_r1m.json 2> $O/prep_r1m.err || exit 1
python3 scripts/ab_libs.py --config recall1m --k 20 --rounds 2 product $B/libcph_base.so | tee $O/ab_recall1m.txt
