#!/bin/bash
# the C2 data at the other bit widths: bench lines with the full-batch parity check against the compiled reference
#     scripts/bits_parity.sh <outdir> <bits> [<bits> ...]
export TMPDIR=/tmp
O=${1:-gpurun_out/bits_parity}; shift; mkdir -p $O
for B in "$@"; do
python3 bench.py --config c2 --bits $B --steps 5 --warmup 1 --no-extra-legs > $O/bench_c2_b$B.json 2> $O/b$B.err; echo "bits $B rc=$?"
python3 -c "
import json; j=json.loads(open('$O/bench_c2_b$B.json').read().strip().splitlines()[-1]); cb=j['cpu_baseline']; r=j['roofline']
print('bits $B: value', round(j['value']), 'kernel frac', round(r['frac'],3), 'exp/q', round(r['expansions_per_query']), 'stream frac', round(j['fastscan_stream']['roofline']['frac'],3), 'ref qps', round(cb['value']), cb['parity_vs_reference'], cb['parity_vs_oracle_counters'])"
done
