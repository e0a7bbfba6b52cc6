R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run12; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_modules.py -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-traffic > $O/bench_$tag.json 2> $O/err_$tag.txt
  python - <<PY
import json
d=json.load(open("$O/bench_$tag.json"))
print("$tag", d["value"], d["ms_per_step"], d["compression"]["ms_per_step"], d["fwd_ms_per_image"])
PY
}
run nocache HIFIC_PACK_CACHE=0
run cache HIFIC_PACK_CACHE=1
run nocache2 HIFIC_PACK_CACHE=0
run cache2 HIFIC_PACK_CACHE=1
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python $R/bench.py --config compression --steps 5 --warmup 2 --no-extras > /tmp/ks.log 2>&1
db=$(find /tmp/ks -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_stats.py $db > $O/kernel_stats_compression.md 2>&1
grep -E "pack|adam" $O/kernel_stats_compression.md | cut -c1-150
