"""Per-kernel, per-launch averages of rocprofv3 --pmc runs as JSON (the profiles/rNN_pmc_*.json files):
python tools/pmc_to_json.py <out.json> "<note>" <dir> [<dir> ...]
hbm_read_bytes_corrected = 2 x FETCH_SIZE x 1024 (gfx950: FETCH_SIZE reports half of the bytes of wide coalesced
streaming reads, MI355X_MICROARCH.md section HBM); hbm_write_bytes = WRITE_SIZE x 1024."""
import csv, glob, json, os, sys, collections
out, note, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
val = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            val[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k, c in val.items():
    row = {"kernel": k[:120], "launches": max(len(v) for v in c.values())}
    for n, v in sorted(c.items()):
        row[n + ("_KB" if n in ("FETCH_SIZE", "WRITE_SIZE") else "")] = round(sum(v) / len(v), 1)
    if "FETCH_SIZE_KB" in row:
        row["hbm_read_bytes_corrected"] = int(2 * row["FETCH_SIZE_KB"] * 1024)
    if "WRITE_SIZE_KB" in row:
        row["hbm_write_bytes"] = int(row["WRITE_SIZE_KB"] * 1024)
    if dur.get(k):
        row["avg_duration_us_under_pmc"] = round(sum(dur[k]) / len(dur[k]), 2)
    rows.append(row)
rows.sort(key=lambda r: -(r.get("hbm_read_bytes_corrected", 0) + r.get("hbm_write_bytes", 0) + r.get("SQ_INSTS_VALU", 0)))
json.dump({"note": note, "kernels": rows}, open(out, "w"), indent=1)
print("wrote", out, len(rows), "kernels")
