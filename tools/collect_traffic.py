"""HBM traffic per launch of each GEMM kernel class from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
usage: collect_traffic.py <fetch.db> <write.db> <out.json>
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64 B per 128 B request for wide streams, so the read
volume is 2 x FETCH_SIZE (calibrated in profiles/r01_pmc_counters.md on the weight-pack kernel)."""
import json, sqlite3, sys

def kind(name):
    if "wgrad" in name and "finalize" not in name and "reduce" not in name: return "wgrad"
    if "gconv_sp9_kernel<2>" in name: return "gconv128"
    if "gconv_sp9_kernel<1>" in name: return "gconv64"
    if "gconv_kernel<" in name:
        args = name[name.index("<") + 1:name.index(">")].split(",")
        wgm, wgn, wm, wn = [int(a) for a in args[-4:]]
        return {128: "gconv128", 64: "gconv64", 32: "gconv32"}[wgm * wm * 32]
    return None

def per_kind(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name", (counter,)).fetchall()
    out = {}
    for name, n, tot in rows:
        k = kind(name)
        if k:
            a = out.setdefault(k, [0, 0.0]); a[0] += n; a[1] += tot
    return out

f = per_kind(sys.argv[1], "FETCH_SIZE"); w = per_kind(sys.argv[2], "WRITE_SIZE")
res = {"_unit": "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes)", "_detail": {}}
for k in sorted(set(f) | set(w)):
    fn, ft = f.get(k, [0, 0.0]); wn, wt = w.get(k, [0, 0.0])
    rd = 2.0 * ft * 1024 / max(fn, 1); wr = wt * 1024 / max(wn, 1)
    res[k] = rd + wr
    res["_detail"][k] = {"launches": fn, "read_bytes": rd, "write_bytes": wr}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res))
