#!/bin/bash
# round 3: GPU suite, then the recall workload (bench line with parity against the compiled reference) on the new beam layout
set -e
export TMPDIR=/tmp
O=gpurun_out/r3_step2; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 500 python3 bench.py --config recall --steps 5 --warmup 2 > $O/bench_recall.json 2> $O/bench_recall.err || { tail -20 $O/bench_recall.err; exit 1; }
python3 - <<'PY'
import json
j=json.loads(open('gpurun_out/r3_step2/bench_recall.json').read().strip().splitlines()[-1])
print('recall qps', j['value'], 'kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'], j.get('cpu_baseline',{}).get('parity_vs_reference'))
PY
