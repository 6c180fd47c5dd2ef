"""Reads the rocprofv3 database of `tools/prof_c5_train.py` (rocprofv3 --kernel-trace --stats -d DIR -o tr) and prints, for the
last iteration, kernel time by name, the number of launches and the idle gaps.  python tools/train_trace.py DIR/tr_results.db [top]"""
import collections, re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = list(c.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "atlas_normalize_kernel" in r[0]]
# an iteration = from the kernel after the previous optimizer step ... here: between consecutive atlas_normalize launches
a, b = idx[-2], idx[-1]
it = rows[a:b]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|at::native::", "", n)
    return n[:120]
agg = collections.defaultdict(lambda: [0, 0])
for n, s, e in it:
    agg[short(n)][0] += 1; agg[short(n)][1] += e - s
tot = sum(v[1] for v in agg.values())
print(f"launches {len(it)}  kernel time {tot / 1e3:.1f} us  span {(it[-1][2] - it[0][1]) / 1e3:.1f} us (next iteration's first launch excluded)")
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{v[1] / 1e3:9.1f} us {v[0]:4d} x  {n}")
gaps = 0.0; prev = None
for n, s, e in it:
    if prev is not None and s > prev: gaps += s - prev
    prev = max(prev or 0, e)
print(f"idle between kernels {gaps / 1e3:.1f} us")
