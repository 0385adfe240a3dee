"""Per-kernel summary of a rocprofv3 SQLite result (rocpd): python tools/db_stats.py path.db [n]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = c.execute("select name, count(*), avg(end-start), min(end-start), sum(end-start) from kernels group by name order by 5 desc").fetchall()
tot = sum(r[4] for r in rows)
for r in rows[:n]:
    print(f"{r[0][:86]:86s} n={r[1]:4d} avg={r[2]/1e3:9.1f}us min={r[3]/1e3:9.1f} pct={100*r[4]/tot:5.1f}")
