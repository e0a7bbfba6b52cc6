R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run35; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_modules.py tests/test_gpu_golden.py tests/test_gpu_conv.py -q -p no:cacheprovider -x > $O/tests.log 2>&1; tail -3 $O/tests.log
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/bench_$tag.json 2> $O/err_$tag.txt
  python - <<PY
import json
d=json.load(open("$O/bench_$tag.json"))
print("$tag", d["value"], d["ms_per_step"], d["compression"]["ms_per_step"], d["fwd_ms_per_image"])
PY
}
run os0 HIFIC_OPT_STREAM=0
run os1
run os0b HIFIC_OPT_STREAM=0
run os1b
echo done
