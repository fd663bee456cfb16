#!/bin/bash
# round-3 diagnostics of the recall workload: phase timers, spilled-beam / probe line counts, FETCH_SIZE calibration
set -e
export TMPDIR=/tmp
O=gpurun_out/r3_diag1; mkdir -p $O
python3 bench.py --config recall --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_recall.json 2> $O/bench_recall.err
echo "bench done" 
python3 scripts/phase_timers.py --config recall --k 20 --lib build/libcph_timers.so > $O/timers.log 2>&1
echo "timers done"
python3 scripts/phase_timers.py --config recall --k 20 --lib build/libcph_traffic.so > $O/traffic.log 2>&1
echo "traffic done"
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/$O/calib_fetch -o p -- $GRAFT_REPO_ROOT/build/fetch_calib > $GRAFT_REPO_ROOT/$O/calib.log 2>&1 )
( cd /tmp && rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum --kernel-trace -d $GRAFT_REPO_ROOT/$O/calib_req -o p -- $GRAFT_REPO_ROOT/build/fetch_calib > $GRAFT_REPO_ROOT/$O/calib2.log 2>&1 ) || echo "req counters failed"
python3 scripts/pmc_summary.py $O/calib_fetch/p_results.db --kernel scatter4 > $O/calib_summary.txt || true
for k in scatter16 line128 line64; do python3 scripts/pmc_summary.py $O/calib_fetch/p_results.db --kernel $k >> $O/calib_summary.txt || true; done
for k in scatter4 scatter16 line128 line64; do python3 scripts/pmc_summary.py $O/calib_req/p_results.db --kernel $k >> $O/calib_summary.txt || true; done
cat $O/calib_summary.txt
tail -3 $O/timers.log; tail -3 $O/traffic.log
