# builds and runs tools/ubench/fetchcal under the two counter passes; prints counter bytes / algorithmic bytes per kernel
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../.. && pwd)}; O=$R/gpurun_out/fetchcal; mkdir -p $O
cd $R/tools/ubench && hipcc --offload-arch=gfx950 -O3 -o /tmp/fetchcal fetchcal.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fc_$c; timeout 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/fc_$c -o fc -- /tmp/fetchcal > /tmp/fc_$c.log 2>&1
  db=$(find /tmp/fc_$c -name "*.db" | head -1)
  python - <<PY
import sqlite3
con=sqlite3.connect("$db"); cur=con.cursor()
rows=cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
B=256*2**20
print("== $c (counter unit: KB) vs 268435456 algorithmic bytes per kernel")
for k,cn,n,v in rows:
    print("  %-52s n=%d  counter %.0f KB = %.3f x algorithmic (x2: %.3f)" % (k[:52], n, v, v*1024/B, 2*v*1024/B))
PY
done > $O/fetchcal.txt 2>&1
cat $O/fetchcal.txt
