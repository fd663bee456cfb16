#!/bin/bash
# the PMC files bench.py reads for roofline.traffic / moved_frac, one per workload (copy <outdir>/pmc_search.json to
# profiles/r<N>_pmc_search_<config>[_b<bits>]_k<k>.json):   scripts/pmc_all.sh [c4]
export TMPDIR=/tmp
( while true; do sleep 60; echo "[keepalive] $(date +%T)" >> gpurun_out/pmc_all_progress.log; done ) &
KA=$!
bash scripts/pmc_search.sh gpurun_out/pmc_c2 c2 10 0 | tail -3
bash scripts/pmc_search.sh gpurun_out/pmc_recall1m recall1m 20 0 | tail -3
bash scripts/pmc_search.sh gpurun_out/pmc_recall1m_b4 recall1m 500 4 | tail -3
[ "$1" = "c4" ] && bash scripts/pmc_search.sh gpurun_out/pmc_c4 c4 10 0 | tail -3
kill $KA
