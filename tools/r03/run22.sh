# effective shader clock of the residual-block kernel (GRBM_GUI_ACTIVE / wall) with and without its memory instructions:
# is the "additive" cost of the loads a lower DVFS clock or stall time?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run22; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MOPS=fwd HIFIC_SP9_W4=1
for v in base abl3 abl15; do
  lib=""; [ $v != base ] && lib="HIFIC_LIB_PATH=$R/gpurun_ab/libhific_$v.so"
  env $lib timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d /tmp/pc_$v -o pc -- python $R/tools/micro_sp9.py 20 > /tmp/pc_$v.log 2>&1
  env $lib timeout 120 rocprofv3 --kernel-trace -d /tmp/kt_$v -o kt -- python $R/tools/micro_sp9.py 20 > /tmp/kt_$v.log 2>&1
  python - $v <<'PY'
import glob, sqlite3, sys
v = sys.argv[1]
def q(pat, sql):
    for db in glob.glob(pat, recursive=True):
        try:
            return sqlite3.connect(db).cursor().execute(sql).fetchall()
        except Exception as e:
            print("query failed", db, e)
    return []
rows = q(f"/tmp/pc_{v}/**/*.db", "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name")
cnt = {c: a for k, c, n, a in rows if "sp9" in k}
cols = [r[1] for r in q(f"/tmp/kt_{v}/**/*.db", "pragma table_info(kernels)")]
name = "name" if "name" in cols else next((c for c in cols if "name" in c), "name")
dur = q(f"/tmp/kt_{v}/**/*.db", f"select avg(end - start) from kernels where {name} like '%sp9%'")
us = dur[0][0] / 1e3 if dur and dur[0][0] else float("nan")
print(v, "avg us (untraced counters)", round(us, 1), {k: round(x) for k, x in cnt.items()},
      "eff clock GHz ~", round(cnt.get("GRBM_GUI_ACTIVE", 0) / (us * 1e3), 2) if us == us else None)
PY
done 2>&1 | tee $O/clock.log
tail -3 /tmp/pc_base.log
