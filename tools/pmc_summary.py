"""Per-kernel averages of a rocprofv3 --pmc run: python tools/pmc_summary.py <dir> [kernel-name substring]"""
import csv, glob, os, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, c in rows.items():
    if pat in k:
        print(k[:90])
        for n, v in sorted(c.items()):
            print("   %-34s %14.0f  (x%d)" % (n, sum(v) / len(v), len(v)))
