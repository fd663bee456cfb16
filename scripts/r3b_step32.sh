#!/bin/bash
# upper-layer descent with sixteen neighbours per pass (four lanes per vector, two chains per lane): tests, step time, latency
export TMPDIR=/tmp
O=gpurun_out/r3b_step32; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit 1
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || exit 1
for i in 1 2; do
python3 bench.py --steps 40 --warmup 5 --no-extra-legs --no-cpu-baseline > $O/bench_new_$i.json 2> $O/bench_new_$i.err
python3 -c "
import json; j=json.loads(open('$O/bench_new_$i.json').read().strip().splitlines()[-1]); print('new descent: value', round(j['value']), 'ms/step', round(j['ms_per_step'],4), 'qps_serial', round(j['qps_serial']), 'host api', round(j['qps_host_api']))"
CPH_LIB_PATH=$PWD/build/libcph_prev.so python3 bench.py --steps 40 --warmup 5 --no-extra-legs --no-cpu-baseline > $O/bench_prev_$i.json 2> $O/bench_prev_$i.err
python3 -c "
import json; j=json.loads(open('$O/bench_prev_$i.json').read().strip().splitlines()[-1]); print('previous:    value', round(j['value']), 'ms/step', round(j['ms_per_step'],4), 'qps_serial', round(j['qps_serial']), 'host api', round(j['qps_host_api']))"
done
python3 scripts/single_query_latency.py c2 | tee $O/single_new.json
CPH_LIB_PATH=$PWD/build/libcph_prev.so python3 scripts/single_query_latency.py c2 | tee $O/single_prev.json
