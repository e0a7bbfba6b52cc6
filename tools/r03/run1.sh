# r03 run 1: exact-index mode - parity tests + cost A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run1; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_exact_index.py tests/test_gpu_golden.py tests/test_gpu_modules.py -q -s -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
grep -E "exact|flips|bf16 vs|FAILED|Error" $O/tests.log | head -60
for e in 0 1; do
  HIFIC_EXACT_INDEX=$e HIFIC_BENCH_DIAG=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-extras 2> $O/bench_e$e.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('exact=$e', d['value'], d['ms_per_step'])"
  grep "bench diag" $O/bench_e$e.err
done
HIFIC_SIDE_WGRAD=0 HIFIC_BRANCH_STREAMS=0 HIFIC_PROF_DUMP=1 timeout 300 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-traffic > $O/prof_dump.json 2> $O/prof_dump.log
grep -c . $O/prof_dump.log
