#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_gate; mkdir -p $O
CPH_BUILD_VERBOSE=1 python3 bench.py --config recall1m --steps 4 --warmup 3 --no-cpu-baseline > $O/gate.json 2> $O/gate.err; echo rc=$?
grep "\[build\]" $O/gate.err | head -20
python3 - <<'PY'
import json
j=json.loads(open('gpurun_out/r3_gate/gate.json').read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"], j["search_stats"])
PY
