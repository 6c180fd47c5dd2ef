#!/bin/bash
# rocprofv3 kernel table of training iterations at config [4]'s real size (eager, padded route, fused AdamW): gpurun_out/r05_train/
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05_train -o t --output-format csv -- python3 $R/tools/prof_c5_train.py 12 1 1 padded > $R/gpurun_out/r05_train.log 2>&1 && echo "train profile done"
