#!/bin/bash
# tools/rehearse_2rank.sh with four ranks sharing the one GPU (test build, shared-memory communicator)
set -o pipefail
mkdir -p gpurun_out
export MCN_USE_TEST_HOOKS=1 MCN_COMM=shm MCN_BENCH_DEVICE=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 4 --steps 20 --warmup 3 --no-cpu-baseline --no-profile > gpurun_out/bench_4rank.json 2> gpurun_out/bench_4rank.err || { tail -20 gpurun_out/bench_4rank.err; exit 1; }
tail -c 900 gpurun_out/bench_4rank.json
