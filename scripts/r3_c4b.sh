#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_c4b; mkdir -p $O
for st in 2 4; do
  python3 bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline --streams $st > $O/c4_s$st.json 2> $O/c4_s$st.err; echo "rc=$?"
  python3 -c "
import json; j=json.loads(open('$O/c4_s$st.json').read().strip().splitlines()[-1]); print('c4 streams $st: qps', round(j['value']), 'ms/step', round(j['ms_per_step'],3), 'kernel_ms', j['roofline']['kernel_ms'], 'pipelined_frac', round(j['roofline']['pipelined_frac'],4))"
done
python3 bench.py --config c4 --steps 20 --warmup 3 --cpu-queries 500 > $O/bench_c4.json 2> $O/bench_c4.err; echo "rc=$?"
