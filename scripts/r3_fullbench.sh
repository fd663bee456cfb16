#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_fullbench; mkdir -p $O
S=$(date +%s)
python3 bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$? wall=$(( $(date +%s) - S )) s"
tail -5 $O/bench.err | cut -c1-300
tail -c 3500 $O/bench.json
