#!/bin/bash
# run on the GPU box from the repo root: control-flow dry run of bench.py --gpus 2 with both ranks on the one GPU (gloo + host transport)
root=$(pwd); mkdir -p $root/gpurun_out
ZRA_BENCH_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 \
  bench.py --gpus 2 --steps 2 --warmup 1 --size-gib 2 --queries 200000 > $root/gpurun_out/bench_2rank_dry.json 2> $root/gpurun_out/bench_2rank_dry.err < /dev/null
echo rc=$?
tail -c 900 $root/gpurun_out/bench_2rank_dry.json
tail -3 $root/gpurun_out/bench_2rank_dry.err
