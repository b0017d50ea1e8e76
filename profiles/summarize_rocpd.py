"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg / min / max.
usage: python profiles/summarize_rocpd.py <results.db> > profiles/<name>.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute(
    "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3,"
    " max(vgpr_count), max(lds_size) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot:.2f} ms over {sum(r[1] for r in rows)} dispatches")
print(f"{'total_ms':>10} {'pct':>6} {'calls':>7} {'avg_us':>10} {'min_us':>9} {'max_us':>10} {'vgpr':>5} {'lds':>7}  kernel")
for r in rows:
    print(f"{r[2]:10.2f} {100*r[2]/tot:6.2f} {r[1]:7d} {r[3]:10.1f} {r[4]:9.1f} {r[5]:10.1f} {r[6]:5d} {r[7]:7d}  {r[0][:160]}")
