#!/bin/bash
# Kernel-trace profiles of a round: the eager bench (every launch a record, S1 alone on the GPU) and the default bench.
#   bash tools/prof_round.sh [tag]      -> gpurun_out/<tag>_eager/, gpurun_out/<tag>_default/ (+ the bench lines as .json)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
export SN_BENCH_EAGER=1 SN_CLASS_BRANCH_FIRST=0
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_eager -o e --output-format csv -- python3 $R/bench.py --steps 200 --warmup 10 --regions 1 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/${tag}_bench_eager_under_rocprof.json 2> $R/gpurun_out/${tag}_eager.err
echo "eager done"
unset SN_BENCH_EAGER SN_CLASS_BRANCH_FIRST
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_default -o d --output-format csv -- python3 $R/bench.py --steps 200 --warmup 10 --regions 3 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/${tag}_default.err
echo "default done"
