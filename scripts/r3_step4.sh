#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r3_step4; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "heap or beams or search_batch_matches or overflow or rerun or capacity" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python3 bench.py --config recall --steps 3 --warmup 1 --no-cpu-baseline > $O/prep.json 2> $O/prep.err
for w in 8 24; do CPH_WAVES_PER_CU=$w python3 scripts/phase_timers.py --product --config recall --k 20 2>/dev/null | tail -1 | cut -c1-120; done
