R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run22; mkdir -p $O
cd $R
export MN=16 MC=120 MK=60 MH=128
for dbg in 0 32 64 96 2 4 102; do
  echo "convt K60 C120 128->256 DBG=$dbg: $(HIFIC_DBG=$dbg timeout 120 python tools/micro_conv.py convt 30 2>&1 | tail -1)"
done 2>&1 | tee $O/ablate.txt
export MN=16 MC=240 MK=120 MH=64
for dbg in 0 32 64 96; do
  echo "convt K120 C240 64->128 DBG=$dbg: $(HIFIC_DBG=$dbg timeout 120 python tools/micro_conv.py convt 30 2>&1 | tail -1)"
done 2>&1 | tee -a $O/ablate.txt
HIFIC_PROF_DUMP=1 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-traffic > $O/bench.json 2> $O/dump.txt
python tools/prof_layers.py $O/dump.txt 4 > $O/layers.md
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], d["compression"]["ms_per_step"], d["fwd_ms_per_image"])
for k,v in d["roofline"]["per_kernel"].items(): print("   ", k, v["ms_per_step"], v["avg_launch_us"], v["tflops"])
PY
