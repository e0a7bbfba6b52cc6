# A/B: sp9 K-split (8 waves) vs 4 waves; im2col rewrite; per-layer dump of the GAN cycle
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run2; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
for ks in 1 2; do
  HIFIC_SP9_KSPLIT=$ks HIFIC_PROF_DUMP=1 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-traffic > $O/bench_ks$ks.json 2> $O/dump_ks$ks.txt
  python tools/prof_layers.py $O/dump_ks$ks.txt 4 > $O/layers_ks$ks.md
  python - <<PY
import json
d=json.load(open("$O/bench_ks$ks.json"))
print("KSPLIT=$ks", d["value"], d["ms_per_step"], d["compression"]["ms_per_step"], d["fwd_ms_per_image"])
for k,v in d["roofline"]["per_kernel"].items(): print("   ", k, v["ms_per_step"], v["avg_launch_us"], v["tflops"])
PY
done
head -40 $O/layers_ks2.md
echo done
