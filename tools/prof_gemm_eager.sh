#!/bin/bash
# eager bench under rocprofv3; prints the GEMM kernels' median durations per grid size
out=${1:-gpurun_out/prof_eager1}
SN_BENCH_EAGER=1 SN_CLASS_BRANCH_FIRST=0 rocprofv3 --kernel-trace --stats -d $out -o e --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $out.json 2> $out.err || exit 1
python3 - $out <<'PY'
import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1] + '/e_kernel_trace.csv')))
d=collections.defaultdict(list)
for r in rows:
    d[(r['Kernel_Name'][:62], r['Grid_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    if len(v)<50 or 'gemm' not in k[0]: continue
    v.sort(); print(k[0].ljust(62), k[1].rjust(8), len(v), 'median', round(v[len(v)//2],1), 'mean', round(sum(v)/len(v),1))
PY
