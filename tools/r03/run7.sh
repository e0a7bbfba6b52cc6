R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run7; mkdir -p $O; cd $R
timeout 600 python tools/graph_probe.py model_fwd_mask5 model_fwd_mask7 gturn_all cycle_all > $O/probe.log 2>&1; cat $O/probe.log
timeout 600 python -m pytest tests/test_gpu_zz_graph.py tests/test_gpu_two_ranks.py -q -s -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
grep -E "^\s+\[|device-RNG|2 ranks|^FAILED|^ERROR|Error" $O/tests.log | cut -c1-250 | head
for g in 1 0; do
  HIFIC_BENCH_GRAPH=$g HIFIC_BENCH_DIAG=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-extras 2> $O/bench_g$g.err > $O/bench_g$g.json; python -c "import json,sys; d=json.loads(open('$O/bench_g$g.json').read()); print('graph=$g', d['value'], d['ms_per_step'], d['config']['launch'])"
  grep -E "bench diag|capture" $O/bench_g$g.err | head -3
done
HIFIC_BENCH_GRAPH=1 HIFIC_BRANCH_STREAMS=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph=1 nobranch', d['value'], d['ms_per_step'])"
HIFIC_BENCH_GRAPH=1 HIFIC_BRANCH_STREAMS=0 HIFIC_SIDE_WGRAD=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph=1 single-stream', d['value'], d['ms_per_step'])"
