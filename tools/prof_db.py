"""per-kernel totals of a rocprofv3 results .db: python tools/prof_db.py DIR [steps]"""
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + '/*.db')[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
n = sum(r[1] for r in rows)
print("launches", n, "per step", n / steps, "kernel ms per step", tot / steps / 1e3)
print("name,calls,total_us,avg_us,percent")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f"{r[0][:90]},{r[1]},{r[2]:.0f},{r[2] / r[1]:.1f},{100 * r[2] / tot:.1f}")
