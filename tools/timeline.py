"""Timeline view of a rocprofv3 kernel trace (rocpd sqlite): how busy is the GPU, per queue and overall, and where are
the idle gaps.  usage: python tools/timeline.py <results.db> [skip_fraction]
Looks at the last (1 - skip_fraction) of the trace (default 0.5: warm-up excluded)."""
import sqlite3
import sys

db = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
con = sqlite3.connect(db)
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("columns:", cols)
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
rows = cur.execute(f"select start, end, {name_col}, {qcol if qcol else 0} from kernels order by start").fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * skip
rows = [r for r in rows if r[0] >= lo]
span = (max(r[1] for r in rows) - rows[0][0]) / 1e6
# union busy time
busy, cur_end, gaps = 0.0, rows[0][0], []
prev = None
for s, e, n, q in rows:
    if s > cur_end:
        gaps.append((s - cur_end, prev, n))
        busy += 0
        cur_start = s
    if e > cur_end:
        busy += (e - max(s, cur_end))
        cur_end = e
        prev = n
print(f"window {span:.2f} ms, {len(rows)} kernels, GPU busy (any kernel) {busy/1e6:.2f} ms = {100*busy/1e6/span:.1f} %, "
      f"sum of kernel durations {sum(r[1]-r[0] for r in rows)/1e6:.2f} ms")
perq = {}
for s, e, n, q in rows:
    perq.setdefault(q, [0, 0])
    perq[q][0] += e - s; perq[q][1] += 1
for q, (d, c) in sorted(perq.items(), key=lambda kv: -kv[1][0]):
    print(f"  queue {q}: {d/1e6:.2f} ms in {c} kernels")
gaps.sort(key=lambda g: -g[0])
tot_gap = sum(g[0] for g in gaps)
print(f"idle gaps: {len(gaps)}, total {tot_gap/1e6:.2f} ms; largest:")
for g, a, b in gaps[:25]:
    print(f"  {g/1e3:8.1f} us  after `{str(a)[:60]}`  before `{str(b)[:60]}`")
# histogram of gap sizes
import collections
h = collections.Counter()
for g, _, _ in gaps:
    h[min(int(g / 1e3) // 5 * 5, 100)] += g
print("gap time by size bucket (us):", {k: round(v / 1e6, 2) for k, v in sorted(h.items())})
