#!/bin/bash
# bench.py lines of the product and of diagnostic builds on ONE box, interleaved (value, kernel alone, full queue):
#     scripts/ab_bench.sh <outdir> "<bench args>" <rounds> product build/libcph_<variant>.so ...
export TMPDIR=/tmp
O=$1; ARGS=$2; R=$3; shift 3; mkdir -p $O
python3 bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --recall-queries 20 > $O/prep.json 2> $O/prep.err || { tail -5 $O/prep.err; exit 1; }
for r in $(seq 1 $R); do for l in "$@"; do
if [ $l = product ]; then unset CPH_LIB_PATH; else export CPH_LIB_PATH=$PWD/$l; fi
python3 bench.py $ARGS --no-extra-legs --no-cpu-baseline --recall-queries 20 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; fq=r.get('full_queue') or {}
print('$l', 'value', round(j['value']), 'ms/step', round(j['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],3), 'full_queue_ms', fq.get('kernel_ms') and round(fq['kernel_ms'],3), 'new/exp', round(j['search_stats']['new_neighbours']/max(1,j['search_stats']['expansions']),2), 'sets', j['config']['batch_sets'])"
done; done | tee $O/summary.txt
