#!/bin/bash
# PC sampling of the search kernel (dynamic profile by instruction): probe what the box supports, then one run each
# on the C2 and the recall workload.  Output under gpurun_out/r3_pcsample/.
export TMPDIR=/tmp
O=gpurun_out/r3_pcsample; mkdir -p $O
rocprofv3-avail info --pc-sampling > $O/avail.txt 2>&1
cat $O/avail.txt | head -40
python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prep.json 2> $O/prep.err || { tail -5 $O/prep.err; exit 1; }
M=${1:-host_trap}; U=${2:-time}; I=${3:-1}
timeout -k 10 400 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M --pc-sampling-unit $U --pc-sampling-interval $I \
   --kernel-trace --output-format csv -d $O/c2 -o p -- python3 scripts/phase_timers.py --product --config c2 --k 10 --reps 2 > $O/c2.log 2>&1
echo "rc=$?"; tail -5 $O/c2.log; ls -la $O/c2 | head
