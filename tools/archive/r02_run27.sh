R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run27; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py -q -p no:cacheprovider -x > $O/tests.log 2>&1; tail -3 $O/tests.log
run() {
  tag=$1; shift
  env "$@" HIFIC_PROF_DUMP=1 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-traffic > $O/bench_$tag.json 2> $O/dump_$tag.txt
  python tools/prof_layers.py $O/dump_$tag.txt 4 > $O/layers_$tag.md
  python - <<PY
import json
d=json.load(open("$O/bench_$tag.json"))
print("$tag", d["value"], d["ms_per_step"], d["compression"]["ms_per_step"], d["fwd_ms_per_image"])
PY
}
run base HIFIC_NO_TPS_SMALL=1
run new
grep "K320 C320\|K220\|C220" $O/layers_base.md | head -14; echo; grep "K320 C320\|K220\|C220" $O/layers_new.md | head -14
echo done
