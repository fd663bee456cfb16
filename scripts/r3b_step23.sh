#!/bin/bash
# probe-first for the narrow codes too (gate workloads): A/B
export TMPDIR=/tmp
O=gpurun_out/r3b_step23; mkdir -p $O
CPH_LIB_PATH=$PWD/build/libcph_pfn.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "search or beams" > $O/pytest_pfn.log 2>&1; rc=$?; tail -3 $O/pytest_pfn.log; [ $rc -eq 0 ] || exit 1
python3 bench.py --config recall --steps 2 --warmup 1 --no-cpu-baseline > $O/prep_r.json 2> $O/prep_r.err || exit 1
python3 scripts/ab_libs.py --config recall --k 20 --rounds 2 product build/libcph_pfn.so | tee $O/ab_recall.txt
python3 bench.py --config recall1m --steps 2 --warmup 1 --no-cpu-baseline > $O/prep_r1m.json 2> $O/prep_r1m.err || exit 1
python3 scripts/ab_libs.py --config recall1m --k 20 --rounds 2 product build/libcph_pfn.so | tee $O/ab_recall1m.txt
