#!/bin/bash
# rocprofv3 kernel tables of eager steps at configs [3] and [4]:  bash tools/prof_shapes.sh [tag] -> gpurun_out/<tag>_c4/, <tag>_c5/
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp
for s in c4 c5; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_$s -o p --output-format csv -- python3 $R/tools/prof_shape.py $s 10 > $R/gpurun_out/${tag}_${s}_prof.txt 2>&1
  echo "$s done"
done
