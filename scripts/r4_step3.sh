#!/bin/bash
# where do the +17 % of step 2 come from: per-kernel durations (rocprofv3 kernel trace) of the product and of the round-3 library
export TMPDIR=/tmp
O=gpurun_out/r4_step3; mkdir -p $O
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep_c2.json 2> $O/prep_c2.err || { tail -5 $O/prep_c2.err; exit 1; }
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_new -- python3 $GRAFT_REPO_ROOT/scripts/phase_timers.py --product --config c2 --reps 5 > $GRAFT_REPO_ROOT/$O/new.json 2> $GRAFT_REPO_ROOT/$O/new.err
CPH_LIB_PATH=$GRAFT_REPO_ROOT/build/libcph_r3.so rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r3 -- python3 $GRAFT_REPO_ROOT/scripts/phase_timers.py --lib $GRAFT_REPO_ROOT/build/libcph_r3.so --config c2 --reps 5 > $GRAFT_REPO_ROOT/$O/r3.json 2> $GRAFT_REPO_ROOT/$O/r3.err
cd $GRAFT_REPO_ROOT
for d in prof_new prof_r3; do echo $d; f=$(find $O/$d -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200; done
