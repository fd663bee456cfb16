#!/bin/bash
# small-batch launch with a prefetch wave: parity tests, single-query / 32-query latency with and without it
export TMPDIR=/tmp
O=gpurun_out/r3b_step7; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 scripts/micro/gate_shape.hip -o /tmp/gate_shape 2> /dev/null && /tmp/gate_shape 100000 | tee $O/gate_shape_100k.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_builder.py -x -q -m gpu -k "small_batch or graph_quality_100k or 100k" > $O/pytest_small.log 2>&1; rc=$?; tail -3 $O/pytest_small.log; [ $rc -eq 0 ] || exit 1
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || exit 1
for i in 1 2; do
python3 scripts/single_query_latency.py c2 | tee -a $O/latency_help.txt
CPH_NO_HELPER_WAVE=1 python3 scripts/single_query_latency.py c2 | tee -a $O/latency_nohelp.txt
done
python3 bench.py --config recall --steps 2 --warmup 1 --no-cpu-baseline > $O/prep_r.json 2> $O/prep_r.err || exit 1
python3 scripts/single_query_latency.py recall | tee -a $O/latency_help_recall.txt
CPH_NO_HELPER_WAVE=1 python3 scripts/single_query_latency.py recall | tee -a $O/latency_nohelp_recall.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
