R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run36; mkdir -p $O
cd $R
export MN=16 MC=60 MK=120 MH=256 MR=3 MS=2
for t in 384 512 768 1024 1536; do echo "wgrad 60->120 s2 target $t: $(HIFIC_WG_TARGET=$t timeout 120 python tools/micro_conv.py wgrad 30 2>&1 | tail -1)"; done
export MN=16 MC=480 MK=960 MH=32
for t in 512 768 1024; do echo "wgrad 480->960 s2 target $t: $(HIFIC_WG_TARGET=$t timeout 120 python tools/micro_conv.py wgrad 30 2>&1 | tail -1)"; done
unset MN MC MK MH MR MS
for t in 512 768 1024; do
  HIFIC_WG_TARGET=$t HIFIC_SIDE_WGRAD=0 HIFIC_BRANCH_STREAMS=0 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-traffic > $O/bench_$t.json 2> /dev/null
  python - <<PY
import json
d=json.load(open("$O/bench_$t.json"))
print("target $t single-stream", d["value"], d["ms_per_step"], d["compression"]["ms_per_step"], d["roofline"]["per_kernel"]["wgrad_kernel<bf16>"])
PY
done
