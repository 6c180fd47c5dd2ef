#!/bin/bash
# rocprofv3 kernel tables of training iterations at config [4]'s real size: eager over the padded route (every launch a record, the
# one-off launches of the first iteration included) and the captured iteration (train.GraphedTrainIter) - gpurun_out/r06_train*/
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06_train -o t --output-format csv -- python3 $R/tools/prof_c5_train.py 12 1 1 padded > $R/gpurun_out/r06_train.log 2>&1 && echo "train profile done"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06_train_graphed -o t --output-format csv -- python3 $R/tools/prof_c5_train.py 24 1 1 graphed > $R/gpurun_out/r06_train_graphed.log 2>&1 && echo "graphed train profile done"
