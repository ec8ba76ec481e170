#!/bin/bash
# run on the GPU box from the repo root: control-flow dry run of bench.py --gpus 8 with all eight ranks on the ONE GPU (gloo + the host
# transport; RCCL refuses duplicate devices): shard -> all-gather of the frame sizes -> gather to the root -> routed serving with 8 owners.
# Never a number: it exists so that the world-8 control flow has run before an 8-GPU node sees it.
root=$(pwd); mkdir -p $root/gpurun_out
ZRA_BENCH_ONE_GPU=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 8 --steps 1 --warmup 1 --size-gib 1 --queries 100000 > $root/gpurun_out/bench_8rank_dry.json 2> $root/gpurun_out/bench_8rank_dry.err < /dev/null
echo rc=$?
tail -c 1200 $root/gpurun_out/bench_8rank_dry.json
tail -3 $root/gpurun_out/bench_8rank_dry.err
