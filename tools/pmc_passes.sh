#!/bin/bash
# the four PMC passes of the eager bench (run on the GPU box: `bash tools/pmc_passes.sh`)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
export SN_BENCH_EAGER=1 SN_BENCH_BATCHES=4
cd /tmp && export TMPDIR=/tmp
i=0
for g in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $g -d $R/gpurun_out/pmc_p$i -o p -- python3 $R/bench.py --steps 5 --warmup 2 --regions 1 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/pmc_p$i.log 2>&1
  echo "pass $i done"
done
