R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run34; mkdir -p $O
cd $R
HIFIC_PROF_DUMP=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/bench.json 2> $O/dump.txt
python tools/prof_layers.py $O/dump.txt 4 > $O/layers.md
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], d["compression"]["ms_per_step"], d["fwd_ms_per_image"])
for k,v in d["roofline"]["per_kernel"].items(): print("   ", k, v["launches_per_step"], v["ms_per_step"], v["avg_launch_us"], v["tflops"])
PY
grep "K320 C320" $O/layers.md | head -4
