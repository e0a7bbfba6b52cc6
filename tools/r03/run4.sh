# r03 run 4: hipGraph capture (events settled), two-rank test, gradient checks (sign-tie criterion), graph on/off A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run4; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_exact_index.py tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_zz_graph.py -q -s -p no:cacheprovider > $O/tests_a.log 2>&1; tail -3 $O/tests_a.log
grep -E "^\s+\[|^    [A-Za-z_.0-9]+: |worst|2 ranks|device-RNG|^FAILED|^ERROR|rounding" $O/tests_a.log | cut -c1-300 | head -70
for g in 1 0; do
  HIFIC_BENCH_GRAPH=$g HIFIC_BENCH_DIAG=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-extras 2> $O/bench_g$g.err > $O/bench_g$g.json; python -c "import json,sys; d=json.loads(open('$O/bench_g$g.json').read()); print('graph=$g', d['value'], d['ms_per_step'], d['config']['launch'])"
  grep -E "bench diag|capture" $O/bench_g$g.err | head -3
done
timeout 1500 python -m pytest tests/test_gpu_fullsize_backward.py -q -s -p no:cacheprovider > $O/tests_b.log 2>&1; tail -3 $O/tests_b.log
grep -E "^\s+\[|^    [A-Za-z_.0-9]+: |worst|^FAILED|^ERROR|rounding|config 5" $O/tests_b.log | cut -c1-400 | head -60
