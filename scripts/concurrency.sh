#!/bin/bash
# concurrent search() callers: the parity test, then aggregate QPS of 1..32 caller threads (ours and the compiled reference)
export TMPDIR=/tmp
O=${1:-gpurun_out/concurrency}; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "concurrent or search_single" -s > $O/pytest.log 2>&1; rc=$?; grep -E "concurrent search|passed|failed|Error" $O/pytest.log | tail -5; [ $rc -eq 0 ] || { tail -30 $O/pytest.log; exit 1; }
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || { tail -5 $O/prep_c2.err; exit 1; }
python3 scripts/concurrent_search.py c2 | tee $O/concurrent_c2.json
python3 scripts/single_query_latency.py c2 | tee $O/single.json
