R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run8; mkdir -p $O
cd $R
run() {
  tag=$1; shift
  env "$@" HIFIC_PROF_DUMP=1 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-traffic > $O/bench_$tag.json 2> $O/dump_$tag.txt
  python tools/prof_layers.py $O/dump_$tag.txt 4 > $O/layers_$tag.md
  python - <<PY
import json
d=json.load(open("$O/bench_$tag.json"))
print("$tag", d["value"], d["ms_per_step"], d["compression"]["ms_per_step"], d["fwd_ms_per_image"])
for k,v in d["roofline"]["per_kernel"].items(): print("   ", k, v["ms_per_step"], v["avg_launch_us"], v["tflops"])
PY
}
run A HIFIC_NO_BIGSTAGE=1
run B HIFIC_NO_BIGSTAGE=1 HIFIC_GC_BIGTILE_MIN_GRID=100000000
run C HIFIC_NO_BIGSTAGE=0
echo done
