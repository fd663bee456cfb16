#!/bin/bash
# QPS at recall@10 >= 0.95 on the gate workload (Gaussian 1M x 128): sweep of k for one bit width, each point with recall
# (first 10 unique ids of k, against the exact MFMA ground truth), QPS, expansions per query and a bit-level check against
# the compiled reference on a bounded query sample.      scripts/gate_k_sweep.sh <bits> <outdir> <k> [<k> ...]
# (profiles/r4_gate_4bit_k_sweep.md = bits 4, k 20 50 100 200 500 1000)
export TMPDIR=/tmp
BITS=${1:-4}; O=${2:-gpurun_out/gate_k_sweep}; shift 2; mkdir -p $O
for K in "$@"; do
timeout -k 10 500 python3 bench.py --config recall1m --bits $BITS --k $K --steps 2 --warmup 1 --cpu-queries 100 --counter-queries 0 --recall-queries 500 --no-extra-legs > $O/k$K.json 2> $O/k$K.err || { tail -5 $O/k$K.err; echo "k=$K failed"; continue; }
python3 -c "
import json; j=json.loads(open('$O/k$K.json').read().strip().splitlines()[-1]); r=j['roofline']; cb=j.get('cpu_baseline',{})
print('bits $BITS k $K: recall@10', round(j['recall_at_10']['k${K}_dedup'],4), 'qps', round(j['value']), 'exp/q', round(r['expansions_per_query']), 'kernel frac', round(r['frac'],3), 'ref qps', round(cb.get('value',0),1), 'parity', cb.get('parity_vs_reference'))"
done | tee $O/summary.txt
