R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run16; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_exact_index.py tests/test_gpu_modules.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_zz_graph.py -q -s -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log; grep -E "^FAILED|^ERROR|Encoder torch|Encoder, bf16|full size vs oracle\]" $O/tests.log | cut -c1-330
for f in 1 0 1 0; do
  HIFIC_EXACT_FUSED=$f timeout 300 python bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('exact_fused=$f', d['value'], d['ms_per_step'])"
done
