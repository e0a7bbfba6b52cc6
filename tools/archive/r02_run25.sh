R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run25; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py tests/test_gpu_modules.py -q -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
HIFIC_WGPIPE_TS=1 timeout 600 python -m pytest tests/test_gpu_conv.py -q -p no:cacheprovider -k "wgrad or weight or bwd" > $O/tests_ts1.log 2>&1; tail -2 $O/tests_ts1.log
for ts in 1 2; do for sh in 0 1; do echo "wgrad 960x960 TS=$ts SH3=$sh: $(HIFIC_WGPIPE_TS=$ts HIFIC_WGPIPE_SH3=$sh timeout 120 python tools/micro_conv.py wgrad 30 2>&1 | tail -1)"; done; done
run() {
  tag=$1; shift
  env "$@" HIFIC_PROF_DUMP=1 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-traffic > $O/bench_$tag.json 2> $O/dump_$tag.txt
  python tools/prof_layers.py $O/dump_$tag.txt 4 > $O/layers_$tag.md
  python - <<PY
import json
d=json.load(open("$O/bench_$tag.json"))
print("$tag", d["value"], d["ms_per_step"], d["compression"]["ms_per_step"], d["fwd_ms_per_image"])
for k,v in d["roofline"]["per_kernel"].items():
    if "pipe" in k: print("   ", k, v["ms_per_step"], v["avg_launch_us"], v["tflops"])
PY
}
run ts1sh1 HIFIC_WGPIPE_TS=1
run ts2sh1
echo done
