"""What the optimizer waits for: the kernels (queue, start, duration) that run in the last `win` us before each adam_prep_kernel of
the steady-state part of a rocprofv3 kernel trace.  usage: python tools/r05/tail.py <results.db> [win_us]"""
import sqlite3, sys
db = sys.argv[1]; win = float(sys.argv[2]) if len(sys.argv) > 2 else 700.0
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select start, end, name, queue_id from kernels order by start").fetchall()
rows = rows[-2600:]
idx = [i for i, r in enumerate(rows) if r[2].startswith("adam_prep")]
for i in idx[-6:]:
    t = rows[i][0]
    print(f"=== adam_prep at {(t - rows[0][0]) / 1e6:.3f} ms")
    for (s, e, n, q) in rows[max(0, i - 60):i + 1]:
        if t - s < win * 1e3:
            print(f"  q{q} start -{(t - s) / 1e3:7.1f} us  dur {(e - s) / 1e3:7.1f} us  {n[:80]}")
