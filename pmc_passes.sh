#!/bin/bash
# usage: pmc_passes.sh <tag> <layer filter> <op filter>: four separate rocprofv3 --pmc passes over one kbench selection
# (SQ issue/wait, SQ pipes, FETCH_SIZE, WRITE_SIZE) -> gpurun_out/pmc_<tag>_{sq1,sq2,fetch,write}.csv
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {  # name, counters
  rm -rf /tmp/pmcout; mkdir -p /tmp/pmcout
  (cd /tmp && timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmcout -o p -- python3 $R/tools/kbench.py "$LAYER" "$OP" > $R/gpurun_out/pmc_$TAG.log 2>&1) || { tail -5 $R/gpurun_out/pmc_$TAG.log; return 1; }
  f=$(find /tmp/pmcout -name "*counter_collection.csv" | head -1)
  cp "$f" $R/gpurun_out/pmc_${TAG}_$1.csv
}
TAG=$1; LAYER=$2; OP=$3
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" &&
run sq2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" &&
run fetch "FETCH_SIZE" &&
run write "WRITE_SIZE"
