R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run37; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -q -p no:cacheprovider -x > $O/tests.log 2>&1; tail -2 $O/tests.log
for t in 0 768; do
  HIFIC_WG_TARGET=$t HIFIC_PROF_DUMP=1 HIFIC_SIDE_WGRAD=0 HIFIC_BRANCH_STREAMS=0 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-traffic > $O/bench_$t.json 2> $O/dump_$t.txt
  python tools/prof_layers.py $O/dump_$t.txt 4 > $O/layers_$t.md
  python - <<PY
import json
d=json.load(open("$O/bench_$t.json"))
print("target $t single-stream", d["value"], d["ms_per_step"], d["compression"]["ms_per_step"], d["roofline"]["per_kernel"]["wgrad_kernel<bf16>"])
PY
done
grep "wgrad M" $O/layers_0.md | head -16
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams on:', d['value'], d['ms_per_step'], d['compression']['ms_per_step'], d['fwd_ms_per_image'])"
