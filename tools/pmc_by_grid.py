"""HBM bytes per launch of rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, split by grid size (the class-side and the instance-side
launch of one kernel differ in their grids; tools/pmc_to_json.py averages them): python tools/pmc_by_grid.py <dir> [<dir> ...]
read = 2 x FETCH_SIZE x 1024 (the gfx950 correction of MI355X_MICROARCH.md), write = WRITE_SIZE x 1024."""
import collections
import csv
import glob
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:72], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = lambda v: sum(v) / len(v) if v else 0.0  # noqa: E731
rows = [(k, mean(v.get("FETCH_SIZE", [])) * 2 * 1024, mean(v.get("WRITE_SIZE", [])) * 1024, len(v.get("FETCH_SIZE", []))) for k, v in acc.items()]
for k, rd, wr, n in sorted(rows, key=lambda r: -(r[1] + r[2])):
    if n >= 5:
        print("%-74s grid %9s launches %3d  read %7.1f MB  write %7.1f MB" % (k[0], k[1], n, rd / 1e6, wr / 1e6))
