#!/bin/bash
# the metric as worded: QPS at recall@10 >= 0.95 with 4-BIT codes at SIFT1M scale (Gaussian 1M x 128): k sweep
export TMPDIR=/tmp
O=gpurun_out/r4_gate4; mkdir -p $O
for K in 20 50 100 200 500 1000; do
timeout -k 10 500 python3 bench.py --config recall1m --bits 4 --k $K --steps 2 --warmup 1 --cpu-queries 100 --recall-queries 500 --no-extra-legs > $O/k$K.json 2> $O/k$K.err || { tail -5 $O/k$K.err; echo "k=$K failed"; continue; }
python3 -c "
import json; j=json.loads(open('$O/k$K.json').read().strip().splitlines()[-1]); r=j['roofline']; cb=j.get('cpu_baseline',{})
print('k $K: recall@10(first 10 unique of k)', round(j['recall_at_10']['k${K}_dedup'],4), 'qps', round(j['value']), 'exp/q', round(r['expansions_per_query']), 'kernel frac', round(r['frac'],3), 'ref qps', round(cb.get('value',0),1), 'parity', cb.get('parity_vs_reference'))"
done | tee $O/summary.txt
