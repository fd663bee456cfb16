#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r3_step6; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
bash scripts/pmc_search.sh $O/pmc_c2 c2 10 > $O/pmc_c2.log 2>&1; tail -22 $O/pmc_c2.log
python3 scripts/single_query_latency.py c2 > $O/single.json 2> $O/single.err || tail -5 $O/single.err
cat $O/single.json | cut -c1-400
