R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run38; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -q -p no:cacheprovider -x -k "im2col or 7x7 or wgrad or weight" > $O/tests.log 2>&1; tail -2 $O/tests.log
for t in 256 512 768 1024; do
  HIFIC_IM2COL_TARGET=$t HIFIC_SIDE_WGRAD=0 HIFIC_BRANCH_STREAMS=0 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-traffic --no-extras > $O/bench_$t.json 2> /dev/null
  python - <<PY
import json
d=json.load(open("$O/bench_$t.json"))
print("im2col target $t", d["value"], d["ms_per_step"])
PY
done
