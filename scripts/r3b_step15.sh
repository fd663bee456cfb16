#!/bin/bash
# is a long query's latency under load its CU's business or the memory system's?  (CU-masked streams)
export TMPDIR=/tmp
O=gpurun_out/r3b_step15; mkdir -p $O
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || exit 1
timeout -k 10 300 python3 scripts/cu_mask_probe.py c2 2>&1 | tee $O/cu_mask_probe.txt
CPH_EXPRESS_CUS=32 timeout -k 10 300 python3 scripts/cu_mask_probe.py c2 2>&1 | tail -2 | tee $O/cu_mask_probe_32.txt
