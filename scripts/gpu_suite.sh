#!/bin/bash
# the GPU parity suite, then the C2 search kernel of this tree against a frozen library (build/libcph_<ref>.so) on the same box
#     scripts/gpu_suite.sh <outdir> [ref-lib]      (ref-lib: e.g. a build of the previous round's tree, see scripts/ab_build.sh)
export TMPDIR=/tmp
O=${1:-gpurun_out/suite}; REF=${2:-build/libcph_r3.so}; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit 1
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || { tail -5 $O/prep_c2.err; exit 1; }
python3 scripts/ab_libs.py --config c2 --rounds 3 product $REF | tee $O/ab_c2.txt
python3 scripts/ab_libs.py --config c2 --nq 100000 --rounds 2 product $REF | tee $O/ab_c2_100k.txt
