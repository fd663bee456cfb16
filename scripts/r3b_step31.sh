#!/bin/bash
# what does the encode + descent kernel cost a step?  (diagnostic build that reuses the encoded batch)
export TMPDIR=/tmp
O=gpurun_out/r3b_step31; mkdir -p $O
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || exit 1
for i in 1 2; do
CPH_LIB_PATH=$PWD/build/libcph_noenc.so python3 bench.py --steps 40 --warmup 5 --no-extra-legs --no-cpu-baseline > $O/bench_noenc_$i.json 2> $O/bench_noenc_$i.err
python3 -c "
import json; j=json.loads(open('$O/bench_noenc_$i.json').read().strip().splitlines()[-1]); print('encode reused: value', round(j['value']), 'ms/step', round(j['ms_per_step'],4), 'sets', j['config']['batch_sets_trial_ms_per_step'], 'qps_serial', round(j['qps_serial']))"
python3 bench.py --steps 40 --warmup 5 --no-extra-legs --no-cpu-baseline > $O/bench_prod_$i.json 2> $O/bench_prod_$i.err
python3 -c "
import json; j=json.loads(open('$O/bench_prod_$i.json').read().strip().splitlines()[-1]); print('product:       value', round(j['value']), 'ms/step', round(j['ms_per_step'],4), 'sets', j['config']['batch_sets_trial_ms_per_step'], 'qps_serial', round(j['qps_serial']))"
done
