#!/bin/bash
# Memory-side PMC passes of the search kernel (address translation, L1->L2 latency, L2 / fabric requests):
#   scripts/pmc_mem.sh <outdir> [config] [k]
# Separate rocprofv3 --pmc passes with --kernel-trace only (run on the GPU box from the repo root).
set -e
OUT=${1:-gpurun_out/pmc_mem}; CFG=${2:-c2}; K=${3:-10}
export TMPDIR=/tmp
mkdir -p "$OUT"
python3 bench.py --config "$CFG" --steps 2 --warmup 1 --no-cpu-baseline --no-recall-leg > "$OUT/bench_prep.json" 2> "$OUT/bench_prep.err"   # builds the index cache
run() { rocprofv3 --pmc $2 --kernel-trace -d "$OUT/$1" -o p -- python3 scripts/phase_timers.py --product --config "$CFG" --k "$K" > "$OUT/$1.log" 2>&1 || echo "pass $1 failed"; }
run utcl1 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
run lat "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"
run lvl "TCC_EA0_RDREQ_LEVEL_sum TCC_REQ_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH"
dbs=""; for p in utcl1 lat tcc lvl sq; do [ -f "$OUT/$p/p_results.db" ] && dbs="$dbs $OUT/$p/p_results.db"; done
python3 scripts/pmc_summary.py $dbs --kernel search_kernel --stats-log "$OUT/utcl1.log" | tee "$OUT/summary.txt"
