#!/bin/bash
# PMC passes of the final search kernel on C4 (10M x 96)
export TMPDIR=/tmp
O=gpurun_out/r3b_step30; mkdir -p $O
bash scripts/pmc_search.sh $O/pmc_c4 c4 10 > $O/pmc_c4.log 2>&1; tail -22 $O/pmc_c4/summary.txt
