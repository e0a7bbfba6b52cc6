# r03 run 2: new parity tests + bench legs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run2; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_exact_index.py tests/test_gpu_golden.py tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_two_ranks.py tests/test_gpu_fullsize_backward.py -q -s -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
grep -E "^\s+\[|worst|flips|2 ranks|FAILED|Error|rounding" $O/tests.log | cut -c1-400 | head -80
HIFIC_BENCH_DIAG=1 timeout 900 python bench.py --steps 10 --warmup 3 --no-traffic > $O/bench.json 2> $O/bench.err; tail -c 6000 $O/bench.json; tail -5 $O/bench.err
