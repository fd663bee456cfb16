#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r3_step5; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python3 bench.py --config recall --steps 3 --warmup 1 --no-cpu-baseline > $O/prep.json 2> $O/prep.err
python3 scripts/phase_timers.py --product --config recall --k 20 2>/dev/null | tail -1 | cut -c1-120
python3 bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/c2.json 2> $O/c2.err
python3 -c "
import json; j=json.loads(open('$O/c2.json').read().strip().splitlines()[-1]); print('c2 qps', j['value'], 'kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'], 'serial', j['qps_serial'])"
