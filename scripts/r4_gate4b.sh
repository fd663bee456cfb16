#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4_gate4b; mkdir -p $O
for K in 500 600 700; do
timeout -k 10 500 python3 bench.py --config recall1m --bits 4 --k $K --steps 3 --warmup 1 --no-cpu-baseline --recall-queries 1000 --no-extra-legs > $O/k$K.json 2> $O/k$K.err || { tail -5 $O/k$K.err; echo "k=$K failed"; continue; }
python3 -c "
import json; j=json.loads(open('$O/k$K.json').read().strip().splitlines()[-1]); r=j['roofline']
print('k $K: recall@10 (1000 queries)', round(j['recall_at_10']['k${K}_dedup'],4), 'qps', round(j['value']), 'exp/q', round(r['expansions_per_query']), 'kernel frac', round(r['frac'],3), 'slots', j['search_stats']['slots'])"
done | tee $O/summary.txt
