#!/bin/bash
# default bench under rocprofv3; prints per-kernel mean durations of the pipelined (replayed) timed region and of the eager pass
out=${1:-gpurun_out/prof_default}
rocprofv3 --kernel-trace --stats -d $out -o d --output-format csv -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > $out.json 2> $out.err || exit 1
python3 - $out <<'PY'
import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1] + '/d_kernel_trace.csv')))
by=collections.defaultdict(list)
for r in rows: by[(r['Kernel_Name'][:58], r['Grid_Size_X'])].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
tot_p=tot_e=0
for k,v in sorted(by.items(), key=lambda kv:-sum(d for _,d in kv[1])):
    if len(v) < 400: continue
    v.sort(); n=len(v)
    e=v[-200:]; g=v[:-200]; p=g[-200:]
    f=lambda x: sum(d for _,d in x)/max(1,len(x))/1e3
    tot_p+=f(p); tot_e+=f(e)
    print(k[0].ljust(58), k[1].rjust(8), n, 'pipelined', round(f(p),1), 'eager', round(f(e),1))
print('sum per step: pipelined', round(tot_p,1), 'eager', round(tot_e,1))
PY
