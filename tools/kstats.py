"""Per-kernel summary of a rocprofv3 rocpd database: python tools/kstats.py <results.db> [steps]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(db.execute("select name, grid_x, grid_y, grid_z, count(*), sum(end-start), avg(end-start), min(end-start) from kernels "
                       "group by name, grid_x, grid_y, grid_z order by 6 desc"))
tot = sum(r[5] for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{r[4]:5d} x avg {r[6]/1e3:8.1f} us (min {r[7]/1e3:7.1f})  per step {r[5]/steps/1e3:8.1f} us {100*r[5]/tot:5.1f}%  grid {r[1]}x{r[2]}x{r[3]}  {r[0][:70]}")
print("total per step %.1f us" % (tot / steps / 1e3))
