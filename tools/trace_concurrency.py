"""How busy the GPU is in the replayed pipeline, from a rocprofv3 kernel trace of bench.py (csv):
python tools/trace_concurrency.py <dir>/bench_kernel_trace.csv   -> share of the time with a kernel running, average
number of kernels in flight, share of the window per kernel (a window in the middle of the hipGraph replays)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
s1 = [e for e in ev if "assign_screen" in e[2]]
n_replayed = len(s1) // 2 + 8                      # the replays come first, then the instrumented eager pass
t0, t1 = s1[n_replayed // 5][0], s1[4 * n_replayed // 5][0]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
busy, cs, ce = 0, None, None
for s, e, _ in win:
    if ce is None or s > ce:
        if ce is not None:
            busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
tot = t1 - t0
print("window %.2f ms, %d kernels: a kernel is running %.1f %% of the time, %.2f kernels in flight on average" % (
    tot / 1e6, len(win), 100.0 * busy / tot, sum(e - s for s, e, _ in win) / tot))
d = collections.defaultdict(float)
for s, e, n in win:
    d[n[:60]] += e - s
for n, v in sorted(d.items(), key=lambda x: -x[1])[:10]:
    print("  %-62s %5.1f %% of the window" % (n, 100.0 * v / tot))
