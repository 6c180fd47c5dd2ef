#!/bin/bash
# PMC passes over the S1 screen alone (tools/diag_s5.py runs it ~90 times): bash tools/pmc_s5.sh   (on the GPU box)
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for g in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $g -d $R/gpurun_out/pmc_s5_p$i -o p -- python3 $R/tools/diag_s5.py > $R/gpurun_out/pmc_s5_p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_s5_p1 assign_screen5 > $R/gpurun_out/pmc_s5_summary.txt
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_s5_p2 assign_screen5 >> $R/gpurun_out/pmc_s5_summary.txt
cat $R/gpurun_out/pmc_s5_summary.txt
