"""Per-kernel PMC counter averages from a rocprofv3 rocpd sqlite database. usage: rocpd_pmc.py <db> [name-substring]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
filt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
print(cols)
q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"
try:
    rows = cur.execute(q).fetchall()
except Exception as e:
    print("query failed", e); rows = []
for k, c, n, v in rows:
    if filt in k:
        print(f"{k[:60]:60s} {c:32s} n={n} avg={v:.1f}")
