R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run5; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py tests/test_gpu_modules.py -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
for v in "1" "0"; do
  HIFIC_NO_BIGSTAGE=$v HIFIC_PROF_DUMP=1 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-traffic > $O/bench_$v.json 2> $O/dump_$v.txt
  python tools/prof_layers.py $O/dump_$v.txt 4 > $O/layers_$v.md
  python - <<PY
import json
d=json.load(open("$O/bench_$v.json"))
print("NO_BIGSTAGE=$v", d["value"], d["ms_per_step"], d["compression"]["ms_per_step"], d["fwd_ms_per_image"])
for k,v in d["roofline"]["per_kernel"].items(): print("   ", k, v["ms_per_step"], v["avg_launch_us"], v["tflops"])
PY
done
head -45 $O/layers_0.md
echo done
