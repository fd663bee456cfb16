#!/bin/bash
# PMC passes for the MFMA kNN kernel (run on the GPU box from the repo root): scripts/prof_knn.sh <outdir> [n] [dim]
set -e
OUT=${1:-gpurun_out/prof_knn}; N=${2:-262144}; DIM=${3:-128}; KERNEL=${4:-knn_mfma}
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
  --kernel-trace -d "$OUT/pmcA" -o a -- python3 scripts/knn_time.py "$N" "$DIM" > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD \
  --kernel-trace -d "$OUT/pmcB" -o b -- python3 scripts/knn_time.py "$N" "$DIM" > "$OUT/b.log" 2>&1
tail -n 3 "$OUT/a.log"
python3 scripts/pmc_summary.py "$OUT/pmcA/a_results.db" "$OUT/pmcB/b_results.db" --kernel "$KERNEL" | tee "$OUT/summary.txt"
python3 - "$OUT/pmcA/a_results.db" <<'PY' | tee -a "$OUT/summary.txt"
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
for r in con.execute("select * from top_kernels limit 3"): print(r)
PY
