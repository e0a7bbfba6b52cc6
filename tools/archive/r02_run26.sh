R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run26; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/tests.log 2>&1; tail -3 $O/tests.log
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/bench_$tag.json 2> $O/err_$tag.txt
  python - <<PY
import json
d=json.load(open("$O/bench_$tag.json"))
print("$tag", d["value"], d["ms_per_step"], d["compression"]["ms_per_step"], d["fwd_ms_per_image"])
PY
}
run side0 HIFIC_SIDE_WGRAD=0
run side1 HIFIC_SIDE_WGRAD=1
run side0b HIFIC_SIDE_WGRAD=0
run side1b HIFIC_SIDE_WGRAD=1
echo done
