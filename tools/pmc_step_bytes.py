"""Sum FETCH_SIZE / WRITE_SIZE over the kernels of a rocprofv3 --pmc run of bench.py and divide by the number of steps:
python tools/pmc_step_bytes.py <dir with FETCH_SIZE run> <dir with WRITE_SIZE run> <steps incl. warm-up and the instrumented pass>"""
import csv, glob, os, sys, collections
def load(d, name):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                tot[r["Kernel_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]] += 1
    return tot, cnt
fetch, cf = load(sys.argv[1], "FETCH_SIZE")
write, cw = load(sys.argv[2], "WRITE_SIZE")
steps = float(sys.argv[3])
rows = []
for k in set(fetch) | set(write):
    rows.append((fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024, cf.get(k, 0), k))
rows.sort(key=lambda r: -(2 * r[0] + r[1]))
tf = sum(r[0] for r in rows) / steps; tw = sum(r[1] for r in rows) / steps
print("per step: FETCH_SIZE %.1f MB (x2 for 16-byte-per-lane streams: %.1f MB), WRITE_SIZE %.1f MB" % (tf / 1e6, 2 * tf / 1e6, tw / 1e6))
for f, w, c, k in rows[:12]:
    print("  %-70s launches/step %4.1f  fetch %7.1f MB  write %7.1f MB" % (k[:70], c / steps, f / steps / 1e6, w / steps / 1e6))
