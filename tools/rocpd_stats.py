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
# gconv_sp9_kernel instantiations serve the residual-block trunk (256 workgroups at the benchmark shape) and the 220 / 320-channel
# layers (320 / 768 workgroups): also list them per grid, so that the trunk's average launch can be read off (bench.py's
# roofline counts the trunk launches only)
def _size_expr(kind):
    """SQL expression for the launch's total work-items (kind 'grid') / work-items per workgroup (kind 'workgroup')."""
    if f"{kind}_size" in cols:
        return f"{kind}_size"
    xyz = [c for c in (f"{kind}_size_x", f"{kind}_size_y", f"{kind}_size_z") if c in cols] or \
          [c for c in (f"{kind}_x", f"{kind}_y", f"{kind}_z") if c in cols]
    return "(" + " * ".join(xyz) + ")" if len(xyz) == 3 else None
gcol, wcol = _size_expr("grid"), _size_expr("workgroup")
if gcol and wcol:
    try:
        extra = cur.execute(f"select {name_col} || ' [' || ({gcol} / {wcol}) || ' workgroups]', count(*), sum(end-start), "
                            f"avg(end-start), min(end-start), max(end-start) from kernels where {name_col} like '%gconv_sp9_kernel%' "
                            f"group by {name_col}, {gcol} order by sum(end-start) desc").fetchall()
        rows = rows + extra
    except Exception as e:                      # older schema: the per-name table stands alone
        print("per-grid rows unavailable:", e, file=sys.stderr)
else:
    print("per-grid rows unavailable; columns of `kernels`:", cols, file=sys.stderr)
total = sum(r[2] for r in rows if ' workgroups]' not in r[0]) or 1
lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for n, c, s, a, mn, mx in rows:
    short = n if len(n) < 110 else n[:107] + "..."
    lines.append(f"| `{short}` | {c} | {s/1e6:.3f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*s/total:.1f} |")
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
print(out)
