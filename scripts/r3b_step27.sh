#!/bin/bash
# N > 1 control flow of bench.py on one rank over RCCL (the driver's scaling run uses the same code with N = 2, 4, 8)
export TMPDIR=/tmp
O=gpurun_out/r3b_step27; mkdir -p $O
CPH_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python3 bench.py --no-extra-legs --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "rc=$?"
tail -3 $O/bench_dist1.err | cut -c1-200
python3 -c "
import json; j=json.loads(open('$O/bench_dist1.json').read().strip().splitlines()[-1]); print('forced dist, 1 rank: value', round(j['value']), 'ms/step', round(j['ms_per_step'],4), 'n_gpus', j['n_gpus'], 'step', j['config']['step'][:120])"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 10 --warmup 2 --no-extra-legs --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; echo "torchrun rc=$?"
tail -c 300 $O/bench_torchrun1.json
