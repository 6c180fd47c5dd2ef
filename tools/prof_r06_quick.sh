#!/bin/bash
# the eager kernel table of the bench step and of configs [3] / [4] only (a subset of tools/prof_r06.sh): bash tools/prof_r06_quick.sh [tag]
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out
T=${1:-r06q}
cd /tmp && export TMPDIR=/tmp
mkdir -p $O
SN_BENCH_EAGER=1 SN_CLASS_BRANCH_FIRST=0 rocprofv3 --kernel-trace --stats -d $O/${T}_eager -o e --output-format csv -- python3 $R/bench.py --steps 200 --warmup 10 --regions 1 --no-cpu-baseline --no-extra-legs > $O/${T}_bench_eager_under_rocprof.json 2> $O/${T}_eager.err && echo "eager done"
for s in c4 c5; do
  rocprofv3 --kernel-trace --stats -d $O/${T}_$s -o p --output-format csv -- python3 $R/tools/prof_shape.py $s 10 > $O/${T}_${s}_prof.txt 2>&1 && echo "$s done"
done
cd $R
python3 tools/trace_split.py $O/${T}_eager > $O/${T}_bench_eager_split_by_grid.txt
head -16 $O/${T}_bench_eager_split_by_grid.txt | cut -c1-170
for s in c4 c5; do python3 tools/trace_split.py $O/${T}_$s | head -14 | cut -c1-170; done
