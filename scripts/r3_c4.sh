#!/bin/bash
# C4 (10M x 96): bench line, then FETCH / WRITE / SQ counters and the memory-side counters of the search kernel
export TMPDIR=/tmp
O=gpurun_out/r3_c4; mkdir -p $O
python3 bench.py --config c4 --steps 10 --warmup 2 --cpu-queries 500 > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench rc=$?"
tail -c 1500 $O/bench_c4.json
bash scripts/pmc_search.sh $O/pmc c4 10 > $O/pmc.log 2>&1; tail -22 $O/pmc.log
bash scripts/pmc_mem.sh $O/mem c4 10 > $O/mem.log 2>&1; tail -26 $O/mem.log
