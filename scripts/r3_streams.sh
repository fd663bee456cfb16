#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r3_streams2; mkdir -p $O
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep.json 2> $O/prep.err
for cfg in "2 0" "4 3072" "4 2560" "4 3584" "4 4096" "2 0" "4 3072" "3 3072" "3 2048"; do
  set -- $cfg
  python3 bench.py --config c2 --steps 60 --warmup 6 --no-cpu-baseline --no-extra-legs --streams $1 --slots $2 > $O/s$1_$2.json 2> $O/s$1_$2.err
  python3 -c "
import json; j=json.loads(open('$O/s$1_$2.json').read().strip().splitlines()[-1]); print('streams $1 slots $2: qps', round(j['value']), 'ms/step', round(j['ms_per_step'],4))"
done
