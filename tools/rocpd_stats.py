"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table (like --stats CSV).
usage: python tools/rocpd_stats.py <results.db> [out.md]"""
import sqlite3
import sys

db = sys.argv[1]
con = sqlite3.connect(db)
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
total = sum(r[2] for r in rows) or 1
lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for n, c, s, a, mn, mx in rows:
    short = n if len(n) < 110 else n[:107] + "..."
    lines.append(f"| `{short}` | {c} | {s/1e6:.3f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*s/total:.1f} |")
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
print(out)
