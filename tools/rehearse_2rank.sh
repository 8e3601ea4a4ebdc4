#!/bin/bash
# the driver's N > 1 launch line rehearsed on the one-GPU box: two ranks share the GPU and exchange through the test build's
# shared-memory communicator (RCCL refuses a second rank on one device); images/s are the rehearsal transport's, value = null
set -o pipefail
mkdir -p gpurun_out
export MCN_USE_TEST_HOOKS=1 MCN_COMM=shm MCN_BENCH_DEVICE=0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err || { tail -20 gpurun_out/bench_2rank.err; exit 1; }
tail -c 1500 gpurun_out/bench_2rank.json
