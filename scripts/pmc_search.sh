#!/bin/bash
# PMC passes of the search kernel on a bench workload (run on the GPU box from the repo root):
#   scripts/pmc_search.sh <outdir> [config] [k] [bits]
# Separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share one), each with --kernel-trace only; writes
# <outdir>/pmc_search.json (copied to profiles/r4_pmc_search_<config>[_b<bits>]_k<k>.json: what bench.py reads for
# `roofline.traffic` / `moved_frac`) and <outdir>/summary.txt.
set -e
OUT=${1:-gpurun_out/pmc_search}; CFG=${2:-c2}; K=${3:-10}; BITS=${4:-0}
export TMPDIR=/tmp
mkdir -p "$OUT"
python3 bench.py --config "$CFG" --bits "$BITS" --k "$K" --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs --recall-queries 20 > "$OUT/bench_prep.json" 2> "$OUT/bench_prep.err"   # builds the index cache
run() { rocprofv3 --pmc $2 --kernel-trace -d "$OUT/$1" -o p -- python3 scripts/phase_timers.py --product --config "$CFG" --k "$K" --bits "$BITS" > "$OUT/$1.log" 2>&1; }
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sqa "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
run sqb "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_SALU"
python3 scripts/pmc_summary.py "$OUT"/fetch/p_results.db "$OUT"/write/p_results.db "$OUT"/sqa/p_results.db "$OUT"/sqb/p_results.db \
    --kernel search_kernel --stats-log "$OUT/fetch.log" --json "$OUT/pmc_search.json" | tee "$OUT/summary.txt"
