#!/bin/bash
export TMPDIR=/tmp
bash scripts/pmc_search.sh gpurun_out/r4_pmc_c2 c2 10 0 | tail -3
bash scripts/pmc_search.sh gpurun_out/r4_pmc_recall1m recall1m 20 0 | tail -3
bash scripts/pmc_search.sh gpurun_out/r4_pmc_recall1m_b4 recall1m 500 4 | tail -3
