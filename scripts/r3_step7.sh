#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r3_step7; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python3 bench.py --config recall --steps 5 --warmup 2 --no-cpu-baseline > $O/recall.json 2> $O/recall.err
python3 -c "
import json; j=json.loads(open('$O/recall.json').read().strip().splitlines()[-1]); print('recall100k qps', j['value'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['search_stats'])"
python3 bench.py --config recall1m --steps 3 --warmup 1 --cpu-queries 100 > $O/recall1m.json 2> $O/recall1m.err
python3 -c "
import json; j=json.loads(open('$O/recall1m.json').read().strip().splitlines()[-1]); print('recall1m qps', j['value'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['search_stats'], j['cpu_baseline']['value'], j['cpu_baseline']['parity_vs_reference'], j['config']['index_build_s'])"
