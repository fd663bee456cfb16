#!/bin/bash
# gate workload (Gaussian 1M x 128, 2-bit, k = 20): resident query slots against kernel time -- do fewer slots (bitmaps that
# fit the 256-MB infinity cache: 125 KB each) beat occupancy?
export TMPDIR=/tmp
O=gpurun_out/r4_gate_slots; mkdir -p $O
for S in 0 1536 2048 3072 4096; do
python3 bench.py --config recall1m --steps 2 --warmup 1 --no-cpu-baseline --slots $S --streams 2 --recall-queries 100 > $O/slots_$S.json 2> $O/slots_$S.err || { tail -5 $O/slots_$S.err; exit 1; }
python3 -c "
import json; j=json.loads(open('$O/slots_$S.json').read().strip().splitlines()[-1]); r=j['roofline']; print('slots $S: value', round(j['value']), 'ms/step', round(j['ms_per_step'],1), 'kernel_ms', round(r['kernel_ms'],1), 'frac', round(r['frac'],3), 'run slots', j['search_stats']['slots'], 'exp/q', round(r['expansions_per_query']))"
done | tee $O/summary.txt
