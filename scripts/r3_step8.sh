#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r3_step8; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python3 bench.py --config recall --steps 2 --warmup 1 --no-cpu-baseline > $O/prep_r.json 2> $O/prep_r.err
python3 scripts/ab_libs.py --config recall --k 20 --rounds 2 product build/libcph_r2.so | tee $O/ab_recall.txt
python3 bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/c2.json 2> $O/c2.err
grep "batches in flight" $O/c2.err
python3 -c "
import json; j=json.loads(open('$O/c2.json').read().strip().splitlines()[-1]); print('c2 qps', j['value'], 'kernel_ms', j['roofline']['kernel_ms'], j['config']['batch_sets'], j['config']['batch_sets_trial_ms_per_step'])"
