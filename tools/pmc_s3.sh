#!/bin/bash
# PMC pass over the S2+S3 kernel alone (tools/time_graph.py), both forms: bash tools/pmc_s3.sh   (on the GPU box)
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  SN_S3_STREAM=$v SN_ZERO_PADDING=0 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD -d $R/gpurun_out/pmc_s3_v$v -o p -- python3 $R/tools/time_graph.py > $R/gpurun_out/pmc_s3_v$v.log 2>&1
  echo "stream=$v rc=$?"
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_s3_v$v instance_graph
done
