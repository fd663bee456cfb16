#!/bin/bash
# the LDS heap's path walk on the vector unit instead of the scalar unit (-DCPH_VALU_WALK): correctness, A/B
export TMPDIR=/tmp
O=gpurun_out/r3b_step29; mkdir -p $O
CPH_LIB_PATH=$PWD/build/libcph_vw.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "heap or search" > $O/pytest_vw.log 2>&1; rc=$?; tail -3 $O/pytest_vw.log; [ $rc -eq 0 ] || exit 1
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || exit 1
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 3 product build/libcph_vw.so | tee $O/ab_c2.txt
python3 scripts/ab_libs.py --config c2 --k 10 --rounds 3 --nq 100000 product build/libcph_vw.so | tee $O/ab_c2_100k.txt
