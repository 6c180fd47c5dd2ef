"""Per-kernel durations of a rocprofv3 kernel trace, split by grid size (the class-side and the instance-side launch of the same kernel
differ in their grids): python tools/trace_split.py <dir with *_kernel_trace.csv> [name substring]"""
import csv, glob, os, sys, collections
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            key = (r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"))
            acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-72s grid %8s wg %5s  n %5d  median %8.1f us  min %8.1f  total %9.1f ms" % (k[0], k[1], k[2], len(v), v[len(v) // 2], v[0], sum(v) / 1e3))
