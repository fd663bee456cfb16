#!/bin/bash
# 2-bit gate workload: the smallest k that meets recall@10 >= 0.95 (first 10 unique of k)
export TMPDIR=/tmp
O=gpurun_out/r4_gate2; mkdir -p $O
for K in 12 14 16 18 20; do
timeout -k 10 500 python3 bench.py --config recall1m --k $K --steps 2 --warmup 1 --no-cpu-baseline --recall-queries 1000 --no-extra-legs > $O/k$K.json 2> $O/k$K.err || { tail -5 $O/k$K.err; echo "k=$K failed"; continue; }
python3 -c "
import json; j=json.loads(open('$O/k$K.json').read().strip().splitlines()[-1]); r=j['roofline']
print('k $K: recall@10 (1000 queries)', round(j['recall_at_10']['k${K}_dedup'],4), 'qps', round(j['value']), 'exp/q', round(r['expansions_per_query']), 'kernel frac', round(r['frac'],3))"
done | tee $O/summary.txt
