"""Context of the large idle gaps of a rocprofv3 kernel trace: the kernels (queue, start, duration) around each gap > thr us in
the last part of the trace.  usage: python tools/r05/gaps.py <results.db> [thr_us] [last_n_kernels]"""
import sqlite3, sys
db = sys.argv[1]; thr = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select start, end, name, queue_id from kernels order by start").fetchall()
nlast = int(sys.argv[3]) if len(sys.argv) > 3 else 2500        # steady state: the last `nlast` kernels of the trace
rows = rows[-nlast:]
cur_end = rows[0][0]; shown = 0
for i, (s, e, n, q) in enumerate(rows):
    if s > cur_end and (s - cur_end) / 1e3 > thr and shown < 8:
        shown += 1
        print(f"=== gap {(s - cur_end) / 1e3:.0f} us at t={(s - rows[0][0]) / 1e6:.2f} ms")
        for (s2, e2, n2, q2) in rows[max(0, i - 6):i + 5]:
            print(f"  q{q2} start {(s2 - rows[0][0]) / 1e6:9.3f} ms dur {(e2 - s2) / 1e3:7.1f} us  {n2[:90]}")
    if e > cur_end: cur_end = e

tot = 0.0; cur_end = rows[0][0]
for (s_, e_, n_, q_) in rows:
    if s_ > cur_end: tot += (s_ - cur_end)
    if e_ > cur_end: cur_end = e_
print(f"window {(cur_end - rows[0][0]) / 1e6:.2f} ms, idle {tot / 1e6:.2f} ms, kernels {len(rows)}")
